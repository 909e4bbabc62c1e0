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


def test_host_helpers_equal_the_reference(ref, tmp_path):
    """The unused-but-public helpers of audio_utilities.py:14-74,117-151 (API-surface completeness): the
    drop-in's versions return what the imported reference returns, value for value."""
    import scipy.io.wavfile
    from tacotron2_amd import audio_utilities as au
    for f in (0.0, 125.0, 700.0, 1234.5, 7600.0, 11025.0):
        assert au.hz_to_mel(f) == ref.hz_to_mel(f)
        assert au.mel_to_hz(au.hz_to_mel(f)) == ref.mel_to_hz(ref.hz_to_mel(f))
    for n_bin, n_fft in ((0, 1024), (10, 1024), (511, 1024), (37, 2048)):
        assert au.fft_bin_to_hz(n_bin, 22050, n_fft) == ref.fft_bin_to_hz(n_bin, 22050, n_fft)
    for f in (0.0, 999.9, 1000.0, 5512.5, 11025.0, 1e9):
        assert au.hz_to_fft_bin(f, 22050, 1024) == ref.hz_to_fft_bin(f, 22050, 1024)
    # get_signal: 16-bit mono and stereo files, and the sample-rate check (audio_utilities.py:117-151)
    rng = np.random.default_rng(2)
    mono = (rng.standard_normal(3000) * 8000).astype(np.int16)
    stereo = (rng.standard_normal((3000, 2)) * 8000).astype(np.int16)
    for name, data in (("mono.wav", mono), ("stereo.wav", stereo)):
        path = str(tmp_path / name)
        scipy.io.wavfile.write(path, 44100, data)
        assert np.array_equal(au.get_signal(path), ref.get_signal(path))
    path = str(tmp_path / "sr.wav")
    scipy.io.wavfile.write(path, 22050, mono)
    assert np.array_equal(au.get_signal(path, expected_fs=22050), ref.get_signal(path, expected_fs=22050))
    with pytest.raises(Exception):
        ref.get_signal(path)
    with pytest.raises(Exception):
        au.get_signal(path)
