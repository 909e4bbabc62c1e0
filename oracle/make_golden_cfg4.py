#!/usr/bin/env python3
"""Pin BASELINE config 4 at FULL size against the REAL reference  --  TEST INFRASTRUCTURE.

Run in the build container only (needs /root/reference; about 25 minutes of one core, 6 GB):

    python oracle/make_golden_cfg4.py

Imports the reference's audio_utilities.py unmodified and runs
reconstruct_signal_griffin_lim (audio_utilities.py:154-188) on the seeded 10-minute input of
oracle/cfg4_inputs.py (T = 51 680, n_fft 2048, hop 256) for 100 iterations.  The reference
always starts from np.random.randn(L) (line 176); that one draw is substituted by the
chosen start signal, and the run is cut into segments (10, +15, +25, +50 iterations) by
feeding a segment's output back in as the next segment's "draw" -- the arithmetic is the
reference's own, unchanged, and the cuts give snapshots that show how the float32 error
grows with the iteration count.

The 53 MB result cannot be committed, so tests/golden/gl_cfg4_fullsize.npz keeps 25 windows
of 8 192 samples (both ends, the 7 seams of the 8-chunk split, 15 spread between) at every
snapshot, the peak, per-block RMS and one random-sign projection per 256-sample block over the whole
signal (cfg4_inputs.block_projections: error energy of every block, not only of the windows) and the
spectral convergence.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")

import audio_utilities as ref  # noqa: E402  (the reference, unmodified)
import cfg4_inputs as c4  # noqa: E402
import gl_oracle as orc  # noqa: E402


def run_segment(mag, x_start, iters):
    saved = np.random.randn
    try:
        np.random.randn = lambda n: np.array(x_start, dtype=np.float64, copy=True)
        return ref.reconstruct_signal_griffin_lim(mag, c4.N_FFT, c4.HOP, iters)
    finally:
        np.random.randn = saved


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else c4.FRAMES      # smaller T: dry run of the script
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "tests", "golden", "gl_cfg4_fullsize.npz")
    fb = orc.mel_filterbank_slaney(22050, c4.N_FFT, 80, 125, 7600)
    mag = c4.magnitudes(fb, frames)
    x = c4.start_signal(frames).astype(np.float64)
    starts = c4.window_starts(frames)
    idx = starts[:, None] + np.arange(c4.WIN)[None, :]
    arrs = {"starts": starts, "frames": frames, "fft_size": c4.N_FFT, "hop": c4.HOP,
            "mag_checksum": np.float64(mag.astype(np.float64).sum()), "x0_checksum": np.float64(x.sum())}
    done = 0
    t0 = time.time()
    for K in c4.SNAP_ITERS:
        x = run_segment(mag, x, K - done)
        done = K
        arrs[f"win{K}"] = x[idx].astype(np.float32)
        arrs[f"peak{K}"] = np.float64(np.max(np.abs(x)))
        print(f"  reference after {K} iterations: peak {arrs[f'peak{K}']:.6f}  ({time.time() - t0:.0f} s)", flush=True)
    nblk = len(x) // 65536
    arrs["block_rms"] = np.sqrt(np.mean(x[:nblk * 65536].reshape(nblk, 65536) ** 2, axis=1))
    arrs["proj100"] = c4.block_projections(x)
    arrs["sc"] = np.float64(orc.spectral_convergence(x, mag, c4.N_FFT, c4.HOP))
    print(f"  spectral convergence {arrs['sc']:.6f}", flush=True)
    # Yardstick: the same algorithm computed by numpy / pocketfft in float32 (gl_oracle.griffin_lim_vec_f32, 4 more
    # minutes).  Ten minutes of audio contain a few spots where Griffin-Lim is locally unstable and float32 rounding
    # grows to several 1e-3 of the peak within 100 iterations, whoever does the arithmetic; its block projections
    # let the GPU test tell "float32 cannot do better here" from "the kernel is wrong".
    y32 = orc.griffin_lim_vec_f32(mag, c4.N_FFT, c4.HOP, c4.ITERS, c4.start_signal(frames))
    arrs["proj100_f32"] = c4.block_projections(y32)
    e32 = np.abs(arrs["proj100_f32"] - arrs["proj100"]) / np.sqrt(c4.PROJ_BLOCK) / arrs["peak100"]
    print(f"  numpy float32 block RMS error / peak: median {np.median(e32):.2e} max {e32.max():.2e}, "
          f"{int((e32 > 1e-3).sum())} blocks above 1e-3", flush=True)
    np.savez_compressed(out, **arrs)
    print(f"wrote {out} ({os.path.getsize(out) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
