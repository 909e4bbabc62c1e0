"""Tier 1: the restatement against the imported reference.  Only runs where
/root/reference exists (the build container); skipped on the GPU box."""
import os
import sys

import numpy as np
import pytest

import gl_oracle as orc

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "audio_utilities.py")),
                                reason="reference checkout not present")


@pytest.fixture(scope="module")
def ref():
    sys.path.insert(0, REF)
    try:
        import importlib
        mod = importlib.import_module("audio_utilities")
        if not mod.__file__.startswith(REF):   # the drop-in shim shadowed it
            import importlib.util
            spec = importlib.util.spec_from_file_location("ref_audio_utilities",
                                                          os.path.join(REF, "audio_utilities.py"))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
        return mod
    finally:
        sys.path.remove(REF)


@pytest.mark.parametrize("N,H,T", [(128, 32, 11), (512, 137, 9), (2048, 256, 6)])
def test_loop_forms(ref, N, H, T):
    rng = np.random.default_rng(N + H)
    x = rng.standard_normal(T * H + N)
    S = ref.stft_for_reconstruction(x, N, H)
    assert np.array_equal(orc.stft_loop(x, N, H), S)
    X = rng.standard_normal(S.shape) + 1j * rng.standard_normal(S.shape)
    assert np.array_equal(orc.istft_loop(X, N, H), ref.istft_for_reconstruction(X, N, H))
    mag = np.abs(rng.standard_normal(S.shape)).astype(np.float32)
    np.random.seed(7)
    y_ref = ref.reconstruct_signal_griffin_lim(mag, N, H, 4)
    np.random.seed(7)
    y = orc.griffin_lim_loop(mag, N, H, 4)
    assert np.array_equal(y, y_ref)
