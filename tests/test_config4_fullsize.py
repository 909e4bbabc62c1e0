"""BASELINE config 4 at its real size against the reference: one 10-minute utterance, T = 51 680 frames,
n_fft 2048 / hop 256, 100 iterations.

tests/golden/gl_cfg4_fullsize.npz was produced by the REAL reference (oracle/make_golden_cfg4.py imports
/root/reference/audio_utilities.py and runs reconstruct_signal_griffin_lim, lines 154-188, on the seeded input
of oracle/cfg4_inputs.py).  The 53 MB result is kept as 25 windows of 8 192 samples -- both ends, the seven
seams of the 8-chunk split, 15 spread between -- after 10, 25, 50 and 100 iterations, plus the peak, the
spectral convergence, and per-block RMS plus one random-sign projection per 256 samples of the whole signal.  Here the same input is regenerated and run
(i) in one piece and (ii) as 8 frame chunks with the exact halo schedule (refresh every 10 iterations), and
both are held to the north-star tolerance against the fixture: max-abs error <= 1e-3 of the peak on every
window, spectral-convergence delta <= 1e-3, and the per-block error bound of check_every_block over the whole signal.
"""
import os

import numpy as np
import pytest
import torch

import cfg4_inputs as c4
import gl_oracle as orc
from tacotron2_amd import audio_utilities as au
from tacotron2_amd import filters
from tacotron2_amd.longform import reconstruct_signal_griffin_lim_longform

pytestmark = pytest.mark.gpu

TOL = 1e-3          # of the waveform peak (north_star), float32 kernels vs the float64 reference
TOL_SC = 1e-3


@pytest.fixture(scope="module")
def cfg4(golden):
    d = golden("gl_cfg4_fullsize")
    assert int(d["frames"]) == c4.FRAMES
    fb = filters.mel(22050, c4.N_FFT, 80, 125, 7600)
    mag = c4.magnitudes(fb)
    x0 = c4.start_signal()
    # the regenerated input is the one the reference saw
    assert float(x0.astype(np.float64).sum()) == float(d["x0_checksum"])
    assert abs(float(mag.astype(np.float64).sum()) / float(d["mag_checksum"]) - 1.0) <= 1e-9
    dev = torch.device("cuda", 0)
    return d, mag, torch.from_numpy(mag).to(dev), torch.from_numpy(x0).to(dev)


def window_error(y, d, K):
    """max |y - reference| over the committed windows, relative to the reference's peak after K iterations."""
    idx = d["starts"][:, None] + np.arange(c4.WIN)[None, :]
    err = np.abs(y[idx].astype(np.float64) - d[f"win{K}"].astype(np.float64))
    return float(err.max() / float(d[f"peak{K}"])), err.max(axis=1) / float(d[f"peak{K}"])


def clusters(blocks, gap=64):
    """Group sorted block indices into spots: runs whose neighbours are at most `gap` blocks apart."""
    out = []
    for b in blocks:
        if out and b - out[-1][1] <= gap:
            out[-1][1] = b
        else:
            out.append([b, b])
    return out


def check_every_block(y, d, what):
    """Every block of the signal, not only the windows: one random-sign projection per 256 samples estimates the
    block's RMS error against the reference (|difference| / 16).

    Ten minutes of audio contain a few spots where Griffin-Lim is locally unstable: float32 rounding grows there to
    several 1e-3 of the peak within 100 iterations WHOEVER does the arithmetic -- numpy's own float32 run of the
    reference algorithm (`proj100_f32`, oracle/gl_oracle.griffin_lim_vec_f32) has 13 of 51 688 blocks above 1e-3 in 3
    spots and a worst block at 6.1e-3; the HIP kernels measured 14 .. 17 blocks, worst 5.8e-3 .. 6.0e-3, mostly at
    the same spots but not only (which of the unstable spots a run excites depends on its rounding: one kernel
    revision woke one that numpy's run left alone).  Everywhere else the error is 1e-7 .. 1e-5.  So the bound that
    is true, and that a wrong kernel (errors everywhere, or at every run seam) cannot meet: median within 1e-5,
    99.9 % of the blocks within 1e-3, the blocks above it few (no more than twice numpy's, plus two) and gathered
    in a handful of short spots, none worse than twice numpy's worst."""
    peak = float(d["peak100"])
    e = np.abs(c4.block_projections(y) - d["proj100"]) / np.sqrt(c4.PROJ_BLOCK) / peak
    e32 = np.abs(d["proj100_f32"] - d["proj100"]) / np.sqrt(c4.PROJ_BLOCK) / peak
    bad, bad32 = np.flatnonzero(e > TOL), np.flatnonzero(e32 > TOL)
    spots, spots32 = clusters(bad), clusters(bad32)
    print("%s: block RMS error / peak from the projections: median %.2e p99 %.2e p99.9 %.2e max %.2e, %d blocks above "
          "1e-3 in %d spots (numpy float32: median %.2e p99.9 %.2e max %.2e, %d blocks in %d spots)" % (
              what, np.median(e), np.quantile(e, 0.99), np.quantile(e, 0.999), e.max(), len(bad), len(spots),
              np.median(e32), np.quantile(e32, 0.999), e32.max(), len(bad32), len(spots32)))
    assert np.median(e) <= 1e-5
    assert np.quantile(e, 0.999) <= TOL
    assert len(bad) <= 2 * len(bad32) + 2
    assert e.max() <= 2.0 * e32.max()
    assert len(spots) <= 2 * len(spots32) + 2
    assert all(hi - lo <= 128 for lo, hi in spots)         # short: 32 k samples = 1.5 s at most


def test_config4_single_piece_against_reference(cfg4):
    d, mag, mag_t, x0_t = cfg4
    grow = {}
    for K in c4.SNAP_ITERS:
        y = au.reconstruct_signal_griffin_lim(mag_t, c4.N_FFT, c4.HOP, K, x_init=x0_t)
        assert tuple(y.shape) == (c4.signal_length(),)
        y = y.cpu().numpy()
        grow[K], _ = window_error(y, d, K)
        assert abs(float(np.max(np.abs(y))) / float(d[f"peak{K}"]) - 1.0) <= TOL
    print("config 4 single piece, max error / peak vs the reference by iteration count:", grow)
    for K in c4.SNAP_ITERS:
        assert grow[K] <= TOL, grow
    # whole-signal checks at 100 iterations: last hop is zero, block RMS, spectral convergence
    assert not y[-c4.HOP:].any()
    nblk = len(y) // 65536
    rms = np.sqrt(np.mean(y[:nblk * 65536].astype(np.float64).reshape(nblk, 65536) ** 2, axis=1))
    assert np.max(np.abs(rms / d["block_rms"] - 1.0)) <= TOL
    check_every_block(y, d, "single piece")
    sc = orc.spectral_convergence(y.astype(np.float64), mag, c4.N_FFT, c4.HOP)
    print("spectral convergence: ours %.6f reference %.6f" % (sc, float(d["sc"])))
    assert abs(sc - float(d["sc"])) <= TOL_SC


def test_config4_eight_chunks_exact_schedule_against_reference(cfg4):
    """The 8-chunk split config 4 names, chunks run in turn on this GPU (the per-rank code path of an 8-GPU run
    minus the transport, which tests/test_longform.py covers over gloo): halo 70 frames, refresh every 10
    iterations.  Held to the same tolerance against the reference, seams included (7 of the windows sit on them)."""
    d, mag, mag_t, x0_t = cfg4
    K = c4.ITERS
    y = reconstruct_signal_griffin_lim_longform(mag_t, c4.N_FFT, c4.HOP, K, n_chunks=c4.N_CHUNKS, exchange_every=10,
                                                x_init=x0_t)
    y = y.cpu().numpy()
    worst, per_window = window_error(y, d, K)
    print("config 4, 8 chunks, max error / peak vs the reference: %.3e (seam windows %.3e)" % (
        worst, max(per_window[i] for i, s in enumerate(d["starts"]) if (s + c4.WIN // 2) % c4.HOP == 0 and s > 0)))
    assert worst <= TOL
    check_every_block(y, d, "8 chunks")
    sc = orc.spectral_convergence(y.astype(np.float64), mag, c4.N_FFT, c4.HOP)
    assert abs(sc - float(d["sc"])) <= TOL_SC
    # pinned run length: the chunked result is bit-identical to the single piece (same sums in the same order)
    y1 = au.reconstruct_signal_griffin_lim(mag_t, c4.N_FFT, c4.HOP, 30, x_init=x0_t, run_frames=10)
    y8 = reconstruct_signal_griffin_lim_longform(mag_t, c4.N_FFT, c4.HOP, 30, n_chunks=c4.N_CHUNKS, exchange_every=10,
                                                 x_init=x0_t, run_frames=10)
    assert torch.equal(y1, y8)
