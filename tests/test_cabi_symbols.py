"""The C-ABI library loads and exports every symbol include/gl_hip.h declares.
No GPU compute here."""
import ctypes
import os
import re

import pytest

from tacotron2_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gl_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gl_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_symbols()
    assert len(names) >= 12
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), f"{n} declared in gl_hip.h but not exported"
    # and the Python binding table covers the header one to one
    assert sorted(_lib.SIGNATURES) == names


def test_geometry_helpers():
    lib = _lib.lib()
    assert lib.gl_version() >= 100
    # L = T*H + N  (audio_utilities.py:110)
    assert lib.gl_signal_length(2048, 256, 861) == 222464
    assert lib.gl_signal_length(1024, 275, 641) == 177299
    # frames = len(range(0, len - N, H))  (audio_utilities.py:91-92)
    for N, H, L in [(2048, 256, 222464), (1024, 275, 177299), (64, 16, 64), (64, 16, 65), (64, 16, 80),
                    (64, 16, 81), (64, 16, 10), (1024, 275, 1024 + 275 * 3 + 1)]:
        assert lib.gl_num_frames(N, H, L) == len(range(0, L - N, H)), (N, H, L)


def test_argument_errors_without_gpu():
    lib = _lib.lib()
    h = ctypes.c_void_p()
    assert lib.gl_plan_create(1000, 256, 10, 1, 0, ctypes.byref(h)) == -1     # not a power of two
    assert b"power of two" in lib.gl_last_error()
    assert lib.gl_plan_create(2048, 0, 10, 1, 0, ctypes.byref(h)) == -1      # hop < 1
    assert lib.gl_plan_create(2048, 256, 0, 1, 0, ctypes.byref(h)) == -1     # no frames
    with pytest.raises(ValueError):
        _lib.check(lib.gl_plan_create(32, 8, 10, 1, 0, ctypes.byref(h)), "gl_plan_create")
    assert lib.gl_stft(None, None, 0, None, 1, 1, None) == -1                # NULL plan
    assert lib.gl_run(None, None, None, None, 1, 1, 1, None) == -1
