"""Host-side pieces of the drop-in (no GPU): WAV encode, helpers, filterbank,
loud failure when no GPU is present."""
import os
import wave

import numpy as np
import pytest

import gl_oracle as orc
from conftest import has_gpu
from tacotron2_amd import audio_utilities as au
from tacotron2_amd import filters, griffin_lim, params


def test_pcm16_known_answers(golden):
    d = golden("pcm16_known")
    assert np.array_equal(au.pcm16_from_float(d["x"]), d["pcm"])
    assert np.array_equal(au.pcm16_from_float(d["x"]), orc.pcm16_from_float(d["x"]))
    with pytest.raises(AssertionError):
        au.pcm16_from_float(np.array([0.0, -1.01]))
    # all-zero input: the reference's x /= 0 gives NaN and int(round(nan)) raises ValueError (lines 204-209)
    with pytest.raises(ValueError):
        au.pcm16_from_float(np.zeros(8))


def test_save_audio_to_file_bytes(tmp_path, capsys):
    rng = np.random.default_rng(0)
    x = rng.uniform(-0.7, 0.7, size=5000)
    keep = x.copy()
    out = tmp_path / "a.wav"
    au.save_audio_to_file(x, 22050, outfile=str(out))
    printed = capsys.readouterr().out.split()
    assert float(printed[0]) == np.max(keep) and float(printed[1]) == np.min(keep)   # audio_utilities.py:200-201
    assert np.array_equal(x, keep)                      # not mutated (extension)
    w = wave.open(str(out))
    assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()) == (1, 2, 22050, 5000)
    data = np.frombuffer(w.readframes(5000), dtype="<i2")
    # the reference's per-sample loop, audio_utilities.py:204-210
    ref = keep / np.max(abs(keep)) * 32767.0
    assert np.array_equal(data, np.array([int(round(v)) for v in ref], dtype="<i2"))
    assert os.path.getsize(out) == 44 + 2 * 5000


def test_small_helpers():
    assert au.hz_to_mel(700.0) == pytest.approx(2595 * np.log10(2.0))
    assert au.mel_to_hz(au.hz_to_mel(1234.5)) == pytest.approx(1234.5)
    assert au.fft_bin_to_hz(10, 22050, 1024) == pytest.approx(10 * 22050 / 2048.0)
    assert au.hz_to_fft_bin(1000.0, 22050, 1024) == int(np.round(1000.0 * 2048 / 22050))
    assert au.hz_to_fft_bin(1e9, 22050, 1024) == 1023


def test_filterbank_matches_oracle_restatement():
    for n_fft in (1024, 2048):
        a = filters.mel(22050, n_fft, 80, 125, 7600)
        b = orc.mel_filterbank_slaney(22050, n_fft, 80, 125, 7600)
        assert a.shape == (80, n_fft // 2 + 1) and a.dtype == np.float32
        assert np.array_equal(a, b)
    assert griffin_lim.filterbank.shape == (params.n_mel_channels, params.n_fft // 2 + 1)
    p = filters.pinv_basis(griffin_lim.filterbank)
    assert p.shape == (513, 80)


def test_params_audio_block():
    # reference params.py:39-45
    assert (params.sampling_rate, params.n_mel_channels, params.frame_hop, params.f_min, params.f_max,
            params.n_fft, params.griffin_lim_iters) == (22050, 80, 275, 125, 7600, 1024, 200)


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU failure mode")
def test_fails_loudly_without_gpu():
    mag = np.ones((4, 33), dtype=np.float32)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        au.reconstruct_signal_griffin_lim(mag, 64, 16, 2)
    with pytest.raises(RuntimeError):
        au.stft_for_reconstruction(np.zeros(200), 64, 16)
    with pytest.raises(RuntimeError):
        griffin_lim.save_mel_to_wav(np.ones((80, 5), dtype=np.float32))


def test_fused_kernel_lds_exchange_layouts():
    """The permuted LDS layouts of the fused kernels' four transposes (gl_fft1024.h kE1P/kE1S/kE2Q/kE2S)
    route every value where the plain transposes do, and no access pattern is worse than 2-way
    bank-conflicting (lane-level model in oracle/sim_fast2048.py; no GPU needed)."""
    import os
    import re
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "oracle"))
    import sim_fast2048 as sim
    src = open(os.path.join(root, "tacotron2_amd", "csrc", "gl_fft1024.h")).read()
    m = re.search(r"kE1P = (\d+), kE1S = (\d+);\s*constexpr int kE2Q = (\d+), kE2S = (\d+);", src)
    assert m and tuple(map(int, m.groups())) == (sim.E1P, sim.E1S, sim.E2Q, sim.E2S)   # model == kernel constants
    assert sim.check_exchange_routing()
    worst = sim.exchange_bank_conflicts()
    assert max(worst.values()) <= 2, worst


def test_cpu_worker_of_the_all_cores_baseline_runs():
    """bench.py's all-cores CPU leg starts oracle/cpu_worker.py once per core: it must run stand-alone
    (no torch, no package import) and print the seconds it took."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "oracle", "cpu_worker.py"), "64", "16", "9", "3", "2", "1"],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert float(out.stdout.strip()) > 0.0
