"""Round-3 GPU tests: the plan cache's capacity model, N1's entry points at config-5 size, non-finite padding
under gl_mel_run, RCCL at world size 1, exact-zero bins on the fused n_fft-2048 path."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import gl_oracle as orc
from tacotron2_amd import filters, griffin_lim
from tacotron2_amd import plan as planmod
from tacotron2_amd.plan import Plan, get_plan

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def peak_err(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


def test_plan_cache_is_sized_by_total_frames_not_by_the_product_of_maxima():
    """One long utterance, then a wide batch of short ones, on one geometry (the order the driver's own GPU run has:
    test_config4_fullsize before the config-3 shard).  The shared plan must hold memory for the larger of the two
    needs, not for max(frames) x max(batch) (round 2: ~45 GB for a 1.7 GB problem), and both calls must work."""
    N, H = 2048, 256
    planmod.clear_plans()
    need = []
    for T, B in ((51680, 1), (861, 128)):
        p = Plan(N, H, T, B)
        need.append(p.device_bytes())
        p.close()
    p1 = get_plan(N, H, 51680, 1)
    assert p1.device_bytes() <= 1.01 * need[0]
    p2 = get_plan(N, H, 861, 128)
    assert p2.max_frames == 51680 and p2.max_batch == 128 and p2.max_total_frames == 861 * 128
    assert p2.device_bytes() <= 1.5 * max(need), (p2.device_bytes(), need)
    # both shapes run on it; a call beyond the frame capacity is refused, not served out of bounds
    rng = np.random.default_rng(0)
    for T, B in ((861, 128), (51680, 1)):
        mag = torch.rand((B, T, N // 2 + 1), device="cuda")
        x0 = torch.randn((B, T * H + N), device="cuda")
        y = p2.run(mag, x0, 2)
        i = int(rng.integers(B))
        ref = orc.griffin_lim_vec(mag[i].cpu().numpy().astype(np.float64), N, H, 2, x0[i].cpu().numpy().astype(np.float64))
        assert peak_err(y[i].cpu().numpy(), ref) <= 1e-4
        del mag, x0, y
    with pytest.raises(ValueError, match="capacity"):
        p2.run(torch.rand((3, 51680, N // 2 + 1), device="cuda"), torch.randn((3, 51680 * H + N), device="cuda"), 1)
    p3 = get_plan(N, H, 861, 128)
    assert p3 is p2                                   # no regrow for a shape it already serves
    planmod.clear_plans()


def test_plan_capacity_argument_errors():
    with pytest.raises(ValueError):
        Plan(2048, 256, 10, 4, max_total_frames=41)   # beyond max_frames * max_batch
    p = Plan(1024, 275, 100, 8, max_total_frames=200)
    mag = torch.rand((2, 100, 513), device="cuda")
    x0 = torch.randn((2, 100 * 275 + 1024), device="cuda")
    p.run(mag, x0, 1)                                 # 200 frames: at the capacity
    with pytest.raises(ValueError, match="capacity"):
        p.run(torch.rand((3, 100, 513), device="cuda"), torch.randn((3, 100 * 275 + 1024), device="cuda"), 1)
    p.close()


@pytest.mark.parametrize("bad", [float("inf"), float("nan"), -float("inf")])
@pytest.mark.parametrize("logmel", [False, True])
def test_mel_run_ignores_non_finite_values_in_padding_frames(bad, logmel):
    """Ragged batch through gl_mel_run: whatever the decoder left in the frames past an utterance's end -- inf,
    NaN (exp overflow on garbage) -- those rows must be exact zeros like gl_mel_to_mag + gl_mask_frames writes
    them, or the overlap-add would carry a NaN into the utterance's valid tail one frame further per iteration."""
    N, H, T, B, K = 2048, 256, 70, 3, 6
    lengths = [70, 41, 9]
    rng = np.random.default_rng(8)
    mel = np.abs(rng.standard_normal((B, 80, T))).astype(np.float32) + 0.1
    for b, n in enumerate(lengths):
        mel[b, :, n:] = bad
    mel[1, 3, lengths[1]:] = 7.0                      # a mix of finite and non-finite garbage
    fb = torch.from_numpy(filters.mel(22050, N, 80, 125, 7600)).cuda()
    x0 = torch.from_numpy(rng.standard_normal((B, T * H + N)).astype(np.float32)).cuda()
    p = Plan(N, H, T, B)
    if logmel:                                        # decoder layout (B, T, 80), garbage planted in the log domain
        lm = np.log(np.where(np.isfinite(mel), mel, 1.0)).astype(np.float32)
        lm[~np.isfinite(mel)] = bad
        src = torch.from_numpy(np.ascontiguousarray(lm.transpose(0, 2, 1))).cuda()
        mag = p.logmel_to_mag(fb, src)
    else:
        src = torch.from_numpy(mel).cuda()
        mag = p.mel_to_mag(fb, src)
    p.mask_frames(mag, lengths)
    two = p.run(mag, x0, K)
    one = p.mel_run(fb, src, x0, K, logmel=logmel, lengths=lengths)
    assert torch.isfinite(one).all()
    assert torch.equal(one, two)
    p.close()


def test_n1_entry_points_at_config5_size(tmp_path, monkeypatch):
    """BASELINE config 5 through the entry points it names (infer.py:24-29): a device-resident (32, 1001, 80)
    log-mel batch -> logmel_to_audio (both return_mag forms) and infer_to_wav; two utterances against the float64
    oracle on exp(logmel).T, the WAV files within 34 LSB (1e-3 of full scale) of the oracle's PCM."""
    from bench import synth_mels
    N, H, T, B, K = 1024, 275, 1001, 32, 60
    assert (griffin_lim.params.n_fft, griffin_lim.params.frame_hop) == (N, H)
    logmel = torch.from_numpy(np.ascontiguousarray(np.log(synth_mels(B, T, 55)).transpose(0, 2, 1))).cuda()
    assert tuple(logmel.shape) == (B, T, 80)
    x0 = np.random.default_rng(6).standard_normal((B, T * H + N)).astype(np.float32)
    x, mag = griffin_lim.logmel_to_audio(logmel, iterations=K, x_init=x0, normalize=None, return_mag=True)
    x2, none = griffin_lim.logmel_to_audio(logmel, iterations=K, x_init=x0, normalize=None, return_mag=False)
    assert none is None and tuple(mag.shape) == (B, T, N // 2 + 1)
    assert torch.equal(x, x2)                         # the same samples where both forms exist
    monkeypatch.setattr(griffin_lim.params, "audio_out_path", str(tmp_path))
    paths = griffin_lim.infer_to_wav(logmel, iterations=K, x_init=x0)
    assert len(paths) == B
    import wave
    fb = filters.mel(22050, N, 80, 125, 7600)
    for i in (3, 30):
        # the oracle's own mel inversion in float64 from the float32 log-mel (infer.py:27-28, griffin_lim.py:18-19)
        mel_i = np.exp(logmel[i].cpu().numpy().astype(np.float64)).T
        m64 = np.dot(fb.T.astype(np.float64), mel_i).T
        assert peak_err(mag[i].cpu().numpy(), m64) <= 2e-6
        ref = orc.griffin_lim_vec(m64, N, H, K, x0[i].astype(np.float64))
        assert peak_err(x[i].cpu().numpy(), ref) <= 1e-3, i
        peak = np.max(np.abs(ref))
        ref_pcm = orc.pcm16_from_float(ref / peak if peak > 1 else ref.copy())
        w = wave.open(paths[i])
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate()) == (1, 2, 22050)
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
        assert pcm.shape == ref_pcm.shape
        assert np.max(np.abs(pcm.astype(int) - ref_pcm.astype(int))) <= 34, i


def test_rccl_world_of_one_rank_on_the_real_stack():
    """sharding.reconstruct_sharded(..., gather=True) over the "nccl" backend (RCCL) with device tensors, in a child
    process: init on the device, barrier, all_reduce, the padded all_gather right behind the library's kernels."""
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "rccl_world1.py")], env=env,
                        capture_output=True, text=True, timeout=300)
    assert pr.returncode == 0, pr.stdout[-2000:] + pr.stderr[-2000:]
    rec = json.loads(pr.stdout.strip().splitlines()[-1])
    assert rec["rccl_world1"] and rec["backend"] == "nccl" and rec["finite"]


def test_all_zero_frames_inside_a_live_signal_project_to_phase_zero():
    """audio_utilities.py:181-183: np.angle(0) == 0, so a frame that is exactly zero projects to (mag, 0) in every
    bin.  Live frames around it must not disturb that (the n_fft-2048 kernels plant 2^-52 in sample 0 of every
    windowed frame instead of testing bins; see DESIGN section 2 for the one deviation this leaves: an exactly zero
    bin inside a NON-zero frame -- a float32 accident that the float64 reference does not share -- gives 0)."""
    N, H, T = 2048, 256, 30
    L = T * H + N
    rng = np.random.default_rng(4)
    mag = (np.abs(rng.standard_normal((T, N // 2 + 1))) + 0.5).astype(np.float32)
    x0 = rng.standard_normal(L).astype(np.float32)
    x0[5 * H:5 * H + N + 3 * H] = 0.0                  # frames 5..8 see only zeros, their neighbours do not
    for K in (1, 2):
        ref = orc.griffin_lim_vec(mag.astype(np.float64), N, H, K, x0.astype(np.float64))
        for path, rf in ((0, 9), (0, None), (1, None)):
            p = Plan(N, H, T, 1)
            p.set_path(path)
            y = p.run(torch.from_numpy(mag).cuda()[None], torch.from_numpy(x0).cuda()[None], K, run_frames=rf)
            assert peak_err(y[0].cpu().numpy(), ref) <= 1e-4, (K, path, rf)
            p.close()


def test_window_sum_table_survives_capture_eager_replay_eager():
    """GL_OPT_WSUM_NORMALISE on a plan used both captured and eagerly: a graph replay rebuilds the window-sum table
    for ITS frame count behind the host's back, so after a capture every call rebuilds the table.  Sequence that
    used to normalise with the wrong sums: capture(T1), eager(T2), replay, eager(T2)."""
    N, H, B = 2048, 256, 2
    T1, T2 = 40, 25
    rng = np.random.default_rng(12)
    p = Plan(N, H, T1, B)

    def inputs(T):
        return (torch.from_numpy(np.abs(rng.standard_normal((B, T, N // 2 + 1))).astype(np.float32)).cuda(),
                torch.from_numpy(rng.standard_normal((B, T * H + N)).astype(np.float32)).cuda())

    m1, x1 = inputs(T1)
    m2, x2 = inputs(T2)
    p.set_options(window_sum_normalize=True)          # configuration call outside the capture
    p.set_options()
    out1 = torch.empty_like(x1)
    with p.capture() as g:
        p.run(m1, x1, 2, out=out1, normalize=True)
    first = p.run(m2, x2, 2, normalize=True).clone()
    g.launch()
    torch.cuda.synchronize()
    again = p.run(m2, x2, 2, normalize=True)
    assert torch.equal(first, again)
    ref = orc.griffin_lim_variant(m2[0].cpu().numpy().astype(np.float64), N, H, 2, x2[0].cpu().numpy().astype(np.float64),
                                  normalize=True)
    body = np.max(np.abs(ref[N:-N]))
    assert np.max(np.abs(again[0].cpu().numpy() - ref)[N:-N]) <= 1e-3 * body
    assert torch.equal(out1, p.run(m1, x1, 2, normalize=True))
    g.close()
    p.close()
