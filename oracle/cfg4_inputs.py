"""Seeded inputs of BASELINE config 4 (one 10-minute utterance)  --  TEST INFRASTRUCTURE.

Shared by oracle/make_golden_cfg4.py (which runs the real reference on them in the build
container) and tests/test_config4_fullsize.py (which regenerates them on the GPU box and
compares the HIP result with the committed reference windows).  Only exactly-rounded
float64 operations and numpy's seeded Generator are used after the draw, plus one float64
exp and one float64 matrix product that are rounded to float32 at the end: the float32
magnitudes are the same on every machine up to rare last-bit ties, far below the parity
tolerance (the smoothed input is well conditioned).

Geometry: SURVEY section 8, config 4: n_fft 2048, hop 256, T = 51 680 frames
(L = 13 232 128 samples = 600.1 s at 22 050 Hz), 100 iterations.
"""
import numpy as np

N_FFT = 2048
HOP = 256
FRAMES = 51680
ITERS = 100
N_CHUNKS = 8
WIN = 8192            # samples per committed window
SNAP_ITERS = (10, 25, 50, 100)


def signal_length(frames=FRAMES):
    return frames * HOP + N_FFT


def mel_10min(frames=FRAMES, seed=4):
    """(80, frames) linear-amplitude mel, exp of a smoothed gaussian field (SURVEY 8d generator,
    zero-padded box smoothing instead of np.roll so the two ends are independent)."""
    rng = np.random.default_rng(seed)
    u = rng.normal(-2.0, 1.5, size=(80 + 2, frames + 2))
    u = (u[:, :-2] + u[:, 1:-1] + u[:, 2:]) / 3.0
    u = (u[:-2] + u[1:-1] + u[2:]) / 3.0
    return np.exp(u)                      # float64


def magnitudes(filterbank, frames=FRAMES, seed=4):
    """griffin_lim.py:18-19 on the synthetic mel: np.dot(filterbank.T, mel).T -> (frames, 1025) float32."""
    mel = mel_10min(frames, seed)
    lin = np.dot(np.asarray(filterbank, dtype=np.float64).T, mel)     # (F, T)
    return np.ascontiguousarray(lin.T.astype(np.float32))


def start_signal(frames=FRAMES, seed=44):
    """The start signal both sides use: float32 values (what the GPU holds), drawn once."""
    return np.random.default_rng(seed).standard_normal(signal_length(frames)).astype(np.float32)


def window_starts(frames=FRAMES):
    """First sample of every committed window: both ends of the signal, the 7 seams of the
    8-chunk split (window centred on the seam's first sample) and 15 more spread between them."""
    L = signal_length(frames)
    starts = {0, L - HOP - WIN, L - WIN}
    for k in range(1, N_CHUNKS):
        seam = (frames * k // N_CHUNKS) * HOP
        starts.add(seam - WIN // 2)
    for k in range(15):
        starts.add(int((k + 0.37) * (L - WIN) / 15.0) // 2 * 2)
    return np.array(sorted(starts), dtype=np.int64)


PROJ_BLOCK = 256


def block_projections(x, seed=7):
    """One random-sign projection per block of 256 samples, sum_i s_i x_i with s_i = +-1 from a seeded Generator:
    a whole-signal fingerprint small enough to commit (52 k numbers).  For a difference signal d the projection of
    a block has variance ||d_block||^2, so |proj(ours) - proj(reference)| / 16 estimates the RMS error of EVERY
    block, not only of the committed windows."""
    x = np.asarray(x, dtype=np.float64)
    n = len(x) // PROJ_BLOCK
    signs = np.random.default_rng(seed).integers(0, 2, size=n * PROJ_BLOCK).astype(np.float64) * 2.0 - 1.0
    return (x[:n * PROJ_BLOCK] * signs).reshape(n, PROJ_BLOCK).sum(axis=1)
