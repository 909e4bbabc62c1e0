#!/usr/bin/env python3
"""BASELINE config 4 on one GPU: a 10-minute mel (T = 51 680 frames, n_fft 2048, hop 256), Griffin-Lim
100 iterations.  Times the single-piece loop and the chunked loop (8 "virtual ranks" run in turn on
this GPU, refresh every 10 iterations), and reports the difference between the two waveforms and the
seam error of the approximate (no-exchange, cross-faded) mode as a function of the halo.
Not the headline bench (bench.py); numbers go to DESIGN.md."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=51680)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--chunks", type=int, default=8)
    args = ap.parse_args()
    from tacotron2_amd import filters, longform
    from tacotron2_amd.plan import get_plan
    from bench import synth_mels

    N, H, T, K, C = 2048, 256, args.frames, args.iters, args.chunks
    L = T * H + N
    dev = torch.device("cuda:0")
    plan = get_plan(N, H, T, 1, dev)
    fb = torch.from_numpy(filters.mel(22050, N, 80, 125, 7600)).to(dev)
    mel = torch.from_numpy(synth_mels(1, T, 77)).to(dev)
    mag = plan.mel_to_mag(fb, mel)[0].contiguous()
    x0 = plan.randn((L,), 42)
    out = {"frames": T, "iters": K, "audio_s": L / 22050.0}

    def timed(fn, reps=2):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            y = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps, y

    t_one, y_one = timed(lambda: longform.reconstruct_signal_griffin_lim_longform(mag, N, H, K, x_init=x0))
    out["single_piece_ms"] = t_one * 1e3
    out["single_piece_audio_s_per_s"] = out["audio_s"] / t_one
    peak = float(y_one.abs().max())
    t_ch, y_ch = timed(lambda: longform.reconstruct_signal_griffin_lim_longform(mag, N, H, K, n_chunks=C,
                                                                                exchange_every=10, x_init=x0))
    out["chunked_exact"] = {"chunks": C, "exchange_every": 10, "halo_frames": 70, "ms_all_chunks_on_one_gpu": t_ch * 1e3,
                            "max_abs_diff_vs_single_piece_over_peak": float((y_ch - y_one).abs().max()) / peak}
    approx = []
    for W in (0, 8, 32, 128, 350, 700):
        xf = min(2 * W * H, 4096)
        y = longform.reconstruct_signal_griffin_lim_longform(mag, N, H, K, n_chunks=C, halo_frames=W, exchange_every=0,
                                                             crossfade=xf, x_init=x0)
        d = (y - y_one).abs()
        approx.append({"halo_frames": W, "crossfade": xf, "max_abs_err_over_peak": float(d.max()) / peak,
                       "rms_err_over_rms": float(d.pow(2).mean().sqrt() / y_one.pow(2).mean().sqrt())})
    out["approximate_no_exchange"] = approx
    print(json.dumps(out))


if __name__ == "__main__":
    main()
