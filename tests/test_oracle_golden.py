"""The CPU oracle against the golden vectors captured from the real reference
(oracle/make_golden.py).  Runs anywhere, no GPU."""
import numpy as np
import pytest

import gl_oracle as orc

GEOMS = ["64_16", "256_100", "1024_275", "2048_256"]


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300))


@pytest.mark.parametrize("g", GEOMS)
def test_stft_matches_reference(golden, g):
    d = golden("stft_" + g)
    N, H = int(d["fft_size"]), int(d["hop"])
    S = orc.stft_loop(d["x"], N, H)
    assert S.shape == d["S"].shape and S.dtype == np.complex128
    assert rel(S, d["S"]) <= 1e-12
    assert rel(orc.stft_vec(d["x"], N, H), d["S"]) <= 1e-12
    # frame count rule, audio_utilities.py:91-92
    assert S.shape[0] == len(range(0, len(d["x"]) - N, H))


@pytest.mark.parametrize("g", GEOMS)
def test_istft_matches_reference(golden, g):
    d = golden("istft_" + g)
    N, H = int(d["fft_size"]), int(d["hop"])
    y = orc.istft_loop(d["X"], N, H)
    assert y.shape == d["y"].shape
    assert rel(y, d["y"]) <= 1e-12
    assert rel(orc.istft_vec(d["X"], N, H), d["y"]) <= 1e-12
    assert np.all(y[-H:] == 0.0)            # F4: no frame covers the last hop samples


@pytest.mark.parametrize("name", ["gl_rand_64_16", "gl_rand_1024_275", "gl_rand_2048_256",
                                  "gl_smooth_1024_275", "gl_smooth_2048_256"])
def test_griffin_lim_matches_reference(golden, name):
    d = golden(name)
    N, H = int(d["fft_size"]), int(d["hop"])
    for K in (1, 2, 5):
        y = orc.griffin_lim_loop(d["mag"], N, H, K, x0=d["x0"])
        assert rel(y, d[f"y{K}"]) <= 1e-12
    assert rel(orc.griffin_lim_vec(d["mag"], N, H, 60, d["x0"]), d["y60"]) <= 1e-9


def test_griffin_lim_default_rng_path(golden):
    """x0 comes from numpy's global legacy RNG like audio_utilities.py:176."""
    d = golden("gl_rand_64_16")
    np.random.seed(int(d["seed"]))
    y = orc.griffin_lim_loop(d["mag"], 64, 16, 2)
    assert rel(y, d["y2"]) <= 1e-12
    np.random.seed(int(d["seed"]))
    assert np.array_equal(np.random.randn(len(d["x0"])), d["x0"])


def test_zero_bins(golden):
    d = golden("gl_zero_bins")
    N, H = int(d["fft_size"]), int(d["hop"])
    for K in (1, 3):
        assert rel(orc.griffin_lim_loop(d["mag"], N, H, K, x0=d["x0"]), d[f"y{K}"]) <= 1e-12
    # angle(0) == 0 -> a zero spectrum projects to (mag, 0)
    S = np.zeros((2, 3), dtype=complex)
    assert np.array_equal(orc.project(S, np.ones((2, 3))), np.ones((2, 3), dtype=complex))


def test_pcm16_known_answers(golden):
    d = golden("pcm16_known")
    assert np.array_equal(orc.pcm16_from_float(d["x"]), d["pcm"])
    with pytest.raises(AssertionError):
        orc.pcm16_from_float(np.array([0.0, 1.5]))


def test_pcm16_random_matches_reference_loop(golden):
    """fp32 samples through the oracle's conversion == the reference's per-sample int(round()) loop."""
    d = golden("pcm16_random")
    assert np.array_equal(orc.pcm16_from_float(d["x"].astype(np.float64)), d["pcm"])


def test_infer_glue_fixture(golden):
    """N1 fixture (infer.py:24-29): exp -> transpose -> save_mel_to_wav body, batch of 3."""
    d = golden("infer_glue")
    N, H, K = int(d["fft_size"]), int(d["hop"]), int(d["iters"])
    fb = orc.mel_filterbank_slaney(22050, N, 80, 125, 7600)
    for b in range(d["logmel"].shape[0]):
        mel = np.exp(d["logmel"][b]).T
        mag = orc.mel_to_linear(fb, mel)
        y = orc.griffin_lim_vec(mag, N, H, K, d["x0"][b].astype(np.float64))
        peak = np.max(np.abs(y))
        assert abs(peak - float(d["peak"][b])) <= 1e-6 * peak
        assert np.max(np.abs(y - d["y"][b])) <= 2e-6 * peak
        pcm = orc.pcm16_from_float(orc.peak_normalise(y))
        assert np.max(np.abs(pcm.astype(int) - d["pcm"][b].astype(int))) <= 1


def test_mel_inversion(golden):
    d = golden("mel_inv_2048")
    got = np.dot(d["basis"], d["mel"]).T
    assert np.allclose(got, d["mag"], rtol=1e-6, atol=1e-6)
    assert orc.mel_to_linear(d["basis"].T, d["mel"]).shape == d["mag"].shape


def test_save_mel_to_wav_cfg1(golden):
    """Full config-1 geometry (n_fft 1024, hop 275, T=641, 60 iters): the oracle
    reproduces the captured reference waveform and PCM."""
    d = golden("save_mel_to_wav_cfg1")
    N, H, K = int(d["fft_size"]), int(d["hop"]), int(d["iters"])
    fb = orc.mel_filterbank_slaney(22050, N, 80, 125, 7600)
    mag = orc.mel_to_linear(fb, d["mel"])
    np.random.seed(int(d["seed"]))
    x0 = np.random.randn(mag.shape[0] * H + N)
    y = orc.griffin_lim_vec(mag, N, H, K, x0)
    peak = np.max(np.abs(y))
    assert abs(peak - float(d["peak"])) <= 1e-6 * float(d["peak"])
    assert np.max(np.abs(y - d["y"])) <= 2e-6 * peak     # fixture stored as fp32
    pcm = orc.pcm16_from_float(orc.peak_normalise(y))
    assert np.max(np.abs(pcm.astype(int) - d["pcm"].astype(int))) <= 1


def test_filterbank_self_consistency():
    """librosa parity is unpinned; check shape, dtype, support and Slaney area norm."""
    fb = orc.mel_filterbank_slaney(22050, 1024, 80, 125, 7600)
    assert fb.shape == (80, 513) and fb.dtype == np.float32
    assert np.all(fb >= 0)
    hz = np.linspace(0, 11025, 513)
    assert np.all(fb[:, hz < 125] == 0) and np.all(fb[:, hz > 7600] == 0)
    # each triangle has (approximately) unit area in Hz
    area = fb.sum(axis=1) * (hz[1] - hz[0])
    assert np.all(np.abs(area[5:] - 1.0) < 0.15)
