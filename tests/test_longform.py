"""Long-form (BASELINE config 4) chunking: frame split with halos, periodic halo refresh, stitching.

CPU part: the chunk bookkeeping (tacotron2_amd.longform.LongformDriver) is driven with the float64
oracle as the per-chunk Griffin-Lim step, in one process and over a world-size-2 gloo group, and
must reproduce the single-piece oracle.  GPU part: the product entry point against the
single-piece fused kernel and against the oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gl_oracle as orc  # noqa: E402  (checker only)

from tacotron2_amd import longform  # noqa: E402


def _stitch_np(out, piece, offset, fade_in, fade_out):
    """numpy statement of gl_stitch_add (include/gl_hip.h)."""
    n = piece.numel()
    i = np.arange(n, dtype=np.float64)
    w = np.ones(n)
    if fade_in:
        w = np.where(i < fade_in, (i + 0.5) / fade_in, w)
    if fade_out:
        w = np.where(n - 1 - i < fade_out, w * ((n - 1 - i) + 0.5) / fade_out, w)
    out[offset:offset + n] += torch.from_numpy(w).to(out.dtype) * piece


def _oracle_driver(mag, N, H, chunks, mine, group=None):
    T = mag.shape[0]

    def step(ch, x_local, n):
        y = orc.griffin_lim_vec(mag[ch.la:ch.lb], N, H, n, x_local[0].numpy())
        return torch.from_numpy(y)[None]

    return longform.LongformDriver(N, H, T, chunks, mine, step, _stitch_np, group)


def _case(N, H, T, seed=3):
    rng = np.random.default_rng(seed)
    mag = np.abs(rng.standard_normal((T, N // 2 + 1)))
    x0 = rng.standard_normal(T * H + N)
    return mag, x0


def _run_chunked(mag, x0, N, H, K, C, W, M, crossfade=0):
    chunks = longform.plan_chunks(mag.shape[0], C, W, N, H)
    drv = _oracle_driver(mag, N, H, chunks, list(range(C)))
    for ch in chunks:
        lo, hi = drv.local_range(ch)
        drv.set_signal(ch, torch.from_numpy(x0[lo:hi].copy())[None])
    drv.run(K, M)
    out = torch.zeros(len(x0), dtype=torch.float64)
    drv.assemble(crossfade, out)
    return out.numpy()


def test_dependency_radius_and_schedule():
    assert longform.dependency_radius(2048, 256) == 7
    assert longform.dependency_radius(1024, 275) == 3
    assert longform.dependency_radius(64, 64) == 0
    assert longform.resolve_schedule(100, 2048, 256) == (70, 10, True)
    assert longform.resolve_schedule(100, 2048, 256, exchange_every=0) == (700, 100, True)
    assert longform.resolve_schedule(100, 2048, 256, halo_frames=21) == (21, 3, True)
    assert longform.resolve_schedule(100, 2048, 256, halo_frames=21, exchange_every=0) == (21, 100, False)
    assert longform.resolve_schedule(4, 2048, 256, halo_frames=700) == (700, 4, True)
    with pytest.raises(ValueError):
        longform.resolve_schedule(10, 2048, 256, exchange_every=-1)


def test_plan_chunks_partition_and_limits():
    chunks = longform.plan_chunks(100, 3, 9, 64, 16)
    assert [c.a for c in chunks] == [0, 34, 67] and chunks[-1].b == 100
    assert chunks[0].la == 0 and chunks[1].la == 25 and chunks[1].lb == 76 and chunks[2].lb == 100
    with pytest.raises(ValueError):
        longform.plan_chunks(30, 3, 9, 64, 16)          # halo + R reaches past the neighbour
    with pytest.raises(ValueError):
        longform.plan_chunks(2, 3, 0, 64, 16)


@pytest.mark.parametrize("N,H,T,K,C,M", [(64, 16, 60, 6, 3, 2), (64, 20, 57, 5, 2, 5), (64, 16, 90, 7, 4, 3),
                                         (128, 33, 50, 4, 2, 1)])
def test_chunked_with_refresh_equals_single_piece(N, H, T, K, C, M):
    """W = M*R halo frames, refresh every M iterations: bit-for-bit the single-piece float64 loop
    (the same frames are summed in the same order into every owned sample)."""
    mag, x0 = _case(N, H, T)
    ref = orc.griffin_lim_vec(mag, N, H, K, x0)
    R = longform.dependency_radius(N, H)
    got = _run_chunked(mag, x0, N, H, K, C, M * R, M)
    assert np.array_equal(got, ref)


def test_halo_one_frame_short_is_not_exact():
    """The bound W >= M*R is tight: one frame less and the seam differs."""
    N, H, T, K, C, M = 64, 16, 60, 6, 3, 2
    mag, x0 = _case(N, H, T)
    ref = orc.griffin_lim_vec(mag, N, H, K, x0)
    R = longform.dependency_radius(N, H)
    got = _run_chunked(mag, x0, N, H, K, C, M * R - 1, M)
    assert not np.array_equal(got, ref)


def test_approximate_mode_error_shrinks_with_overlap():
    """No exchanges, halo shorter than K*R, cross-faded seams: the error against the single-piece
    result falls as the overlap grows and vanishes at W = K*R."""
    N, H, T, K, C = 64, 16, 120, 6, 2
    mag, x0 = _case(N, H, T, seed=11)
    ref = orc.griffin_lim_vec(mag, N, H, K, x0)
    R = longform.dependency_radius(N, H)
    errs = []
    for W in (2, 6, 12, K * R):
        got = _run_chunked(mag, x0, N, H, K, C, W, K, crossfade=min(2 * W * H, 4 * H) if W < K * R else 0)
        errs.append(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))
    assert errs[-1] == 0.0
    assert errs[0] > errs[2] > 0.0
    assert errs[1] >= errs[2]


def test_crossfade_weights_sum_to_one():
    out = torch.zeros(100, dtype=torch.float64)
    ones = torch.ones(60, dtype=torch.float64)
    _stitch_np(out, ones, 0, 0, 20)       # covers [0, 60), fades out over [40, 60)
    _stitch_np(out, ones, 40, 20, 0)      # covers [40, 100), fades in over [40, 60)
    assert np.allclose(out.numpy(), 1.0, atol=1e-15)


# ---- world size 2 over gloo -------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        N, H, T, K, M = 64, 16, 70, 7, 2
        mag, x0 = _case(N, H, T, seed=21)
        R = longform.dependency_radius(N, H)
        chunks = longform.plan_chunks(T, world, M * R, N, H)
        drv = _oracle_driver(mag, N, H, chunks, [rank])
        lo, hi = drv.local_range(chunks[rank])
        drv.set_signal(chunks[rank], torch.from_numpy(x0[lo:hi].copy())[None])
        drv.run(K, M)
        out = torch.zeros(len(x0), dtype=torch.float64)
        drv.assemble(0, out)
        q.put((rank, out.numpy()))
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo_halo_exchange_equals_single_piece():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    N, H, T, K = 64, 16, 70, 7
    mag, x0 = _case(N, H, T, seed=21)
    ref = orc.griffin_lim_vec(mag, N, H, K, x0)
    for _, full in results:
        assert np.array_equal(full, ref)


# ---- GPU: product entry point ----------------------------------------------------------------------------
def _smooth_mag(T, F, seed):
    rng = np.random.default_rng(seed)
    u = rng.normal(-2.0, 1.5, size=(T, F))
    for _ in range(2):
        u = (u + np.roll(u, 1, 0) + np.roll(u, -1, 0)) / 3.0
        u = (u + np.roll(u, 1, 1) + np.roll(u, -1, 1)) / 3.0
    return np.exp(u).astype(np.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("N,H,T,K,C,M", [(2048, 256, 400, 12, 3, 4), (1024, 275, 300, 10, 4, 5), (2048, 256, 260, 9, 2, 0)])
def test_gpu_chunked_matches_single_piece_and_oracle(N, H, T, K, C, M):
    """exchange_every M (0 = never, halo K*R): same waveform as the single-piece fused kernel up to
    fp32 summation order at run seams, and within the north-star tolerance of the float64 oracle."""
    from tacotron2_amd import audio_utilities as au
    mag = _smooth_mag(T, N // 2 + 1, 5)
    x0 = np.random.default_rng(6).standard_normal(T * H + N).astype(np.float32)
    one = au.reconstruct_signal_griffin_lim(torch.from_numpy(mag), N, H, K, x_init=torch.from_numpy(x0),
                                            device="cuda:0")
    got = longform.reconstruct_signal_griffin_lim_longform(torch.from_numpy(mag), N, H, K, n_chunks=C,
                                                           exchange_every=M, x_init=torch.from_numpy(x0),
                                                           device="cuda:0")
    one, got = one.cpu().numpy(), got.cpu().numpy()
    peak = np.max(np.abs(one))
    assert np.max(np.abs(got - one)) <= 2e-5 * peak
    ref = orc.griffin_lim_vec(mag.astype(np.float64), N, H, K, x0.astype(np.float64))
    assert np.max(np.abs(got - ref)) <= 1e-3 * np.max(np.abs(ref))           # north_star tolerance


@pytest.mark.gpu
@pytest.mark.parametrize("N,H,T,K,C,M,G", [(2048, 256, 900, 20, 3, 5, 16), (1024, 275, 700, 12, 2, 4, 12)])
def test_gpu_chunked_is_bit_identical_with_pinned_run_length(N, H, T, K, C, M, G):
    """Chunks that start on a multiple of the pinned run length sum every owned sample in the same
    order as the single-piece launch: the refresh schedule is exact in fp32 too, bit for bit.
    The n_fft-1024 kernel transforms two frames in one complex FFT, so a stale halo frame perturbs
    its partner's rounding: bit-identity there needs one more halo frame per iteration (R + 1)."""
    mag = torch.from_numpy(_smooth_mag(T, N // 2 + 1, 8))
    x0 = torch.from_numpy(np.random.default_rng(9).standard_normal(T * H + N).astype(np.float32))
    kw = dict(x_init=x0, device="cuda:0", run_frames=G)
    one = longform.reconstruct_signal_griffin_lim_longform(mag, N, H, K, **kw)
    R = longform.dependency_radius(N, H) + (1 if N == 1024 else 0)
    got = longform.reconstruct_signal_griffin_lim_longform(mag, N, H, K, n_chunks=C, exchange_every=M,
                                                           halo_frames=M * R, **kw)
    assert torch.equal(one, got)


@pytest.mark.gpu
def test_gpu_device_rng_is_split_invariant_and_stitch_matches_numpy():
    from tacotron2_amd.plan import get_plan
    plan = get_plan(2048, 256, 8, 1, "cuda:0")
    full = plan.randn((10007,), 99).cpu().numpy()
    for off, n in [(0, 10007), (1, 50), (4, 64), (4093, 1030), (10006, 1)]:
        part = plan.randn((n,), 99, offset=off).cpu().numpy()
        assert np.array_equal(part, full[off:off + n]), (off, n)
    assert abs(full.mean()) < 0.05 and abs(full.std() - 1.0) < 0.05
    rng = np.random.default_rng(0)
    out0 = rng.standard_normal(5000).astype(np.float32)
    piece = rng.standard_normal(1200).astype(np.float32)
    for off, fi, fo in [(100, 0, 0), (3900, 300, 0), (-50, 0, 400), (0, 600, 600)]:
        out = torch.from_numpy(out0.copy()).cuda()
        plan.stitch_add(out, torch.from_numpy(piece).cuda(), off, fi, fo)
        exp = torch.from_numpy(out0.astype(np.float64))
        lo, hi = max(0, off), min(5000, off + 1200)
        tmp = torch.zeros(1200 + 5000 + 100, dtype=torch.float64)
        _stitch_np(tmp, torch.from_numpy(piece.astype(np.float64)), off + 100, fi, fo)
        exp = exp + tmp[100:5100]
        assert np.allclose(out.cpu().numpy(), exp.numpy(), rtol=0, atol=2e-6), (off, fi, fo)
        assert lo < hi
    with pytest.raises(ValueError):
        plan.stitch_add(torch.zeros(10).cuda(), torch.zeros(4).cuda(), 0, 3, 2)


def _gpu_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        N, H, T, K = 2048, 256, 500, 12
        mag = torch.from_numpy(_smooth_mag(T, N // 2 + 1, 15))
        x0 = torch.from_numpy(np.random.default_rng(16).standard_normal(T * H + N).astype(np.float32))
        y = longform.reconstruct_signal_griffin_lim_longform(mag, N, H, K, exchange_every=4, x_init=x0,
                                                             device="cuda:0", run_frames=16)
        q.put((rank, y.cpu().numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_gpu_two_ranks_over_gloo_equal_single_piece():
    """The one-chunk-per-rank path of the product entry point: two processes share cuda:0, the halo
    refresh and the final gather go through torch.distributed (gloo here, staged through the host;
    RCCL send/recv on a multi-GPU node).  Bit-identical to the single-piece run with the same pin."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    N, H, T, K = 2048, 256, 500, 12
    mag = torch.from_numpy(_smooth_mag(T, N // 2 + 1, 15))
    x0 = torch.from_numpy(np.random.default_rng(16).standard_normal(T * H + N).astype(np.float32))
    one = longform.reconstruct_signal_griffin_lim_longform(mag, N, H, K, x_init=x0, device="cuda:0", run_frames=16)
    one = one.cpu().numpy()
    for _, full in results:
        assert np.array_equal(full, one)
