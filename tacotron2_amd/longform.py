"""Long-form Griffin-Lim: one long utterance cut into frame chunks (BASELINE config 4).

The reference reconstructs an utterance in one piece (audio_utilities.py:154-188).  Griffin-Lim
does not decompose by frames exactly: a frame's inverse transform lands on N samples, which the
next iteration's frames up to R = ceil(N/H) - 1 positions away read back, so information travels
R frames per iteration (7 at n_fft 2048 / hop 256, 3 at 1024 / 275).  The split used here trades
that dependency for redundant frames instead of per-iteration messages:

* chunk c OWNS frames [a, b) and additionally computes a halo of W frames on each interior side
  (local frames [la, lb) = [a - W, b + W) clipped to the utterance);
* after a refresh the local signal is right everywhere; every iteration the outermost R halo
  frames per side go stale (their neighbours are missing) while everything further in stays
  exact, so after M iterations with W >= M*R the owned samples [a*H, b*H) are still exactly the
  single-piece result;
* every M iterations neighbouring chunks swap the edge of their owned samples (W*H + N - H
  floats per side, 72 KB at M = 10) to refresh each other's halo: K/M - 1 point-to-point
  messages per boundary over the neighbour's xGMI link instead of K, at 2*M*R/(b - a) redundant
  work (2 % for a 10-minute utterance on 8 GPUs at M = 10);
* W >= K*R needs no messages at all ("independent overlapping chunks", still exact); a smaller
  halo without exchanges is the approximate mode, whose seams are cross-faded (gl_stitch_add)
  and whose error against the single-piece result is measured in tests/ and DESIGN.md.

One process per GPU: chunk index = rank, exchanges are torch.distributed send/recv (RCCL when the
group's backend is nccl; staged through the host for gloo).  Without a process group the chunks
run one after the other on the current GPU ("virtual ranks"), which is also how the split is
checked against the single-piece kernel on a one-GPU box.
"""
from collections import namedtuple

import numpy as np
import torch
import torch.distributed as dist

from .sharding import shard_bounds

Chunk = namedtuple("Chunk", "index a b la lb")      # frames: owned [a, b), computed [la, lb)


def dependency_radius(n_fft, hop):
    """Frames per iteration that a change travels: ceil(N/H) - 1 (see the module docstring)."""
    return -(-int(n_fft) // int(hop)) - 1


def plan_chunks(n_frames, n_chunks, halo, n_fft, hop, align=1):
    """Contiguous balanced split of `n_frames` into `n_chunks` owned ranges plus `halo` frames of
    redundant work on each interior side.  Halos may only reach into the adjacent chunk.
    align: local range widened to multiples of it (the fused kernels' run length, so that a
    chunk's wave-runs -- and the n_fft-1024 kernel's frame pairs, for an even run length --
    coincide with the single-piece launch's and every sum associates alike)."""
    T, C, W, A = int(n_frames), int(n_chunks), int(halo), max(1, int(align))
    if C < 1 or T < C:
        raise ValueError(f"cannot cut {T} frames into {C} chunks")
    if W < 0:
        raise ValueError("halo must be >= 0")
    R = dependency_radius(n_fft, hop)
    chunks = []
    for c in range(C):
        a, b = shard_bounds(T, C, c)
        chunks.append(Chunk(c, a, b, max(0, a - W) // A * A, min(T, -(-(b + W) // A) * A)))
    if C > 1:
        smallest = min(ch.b - ch.a for ch in chunks)
        if max(max(ch.a - ch.la, ch.lb - ch.b) for ch in chunks) + R > smallest:
            raise ValueError(f"halo {W} + dependency radius {R} frames exceeds the smallest chunk ({smallest} "
                             f"frames): use fewer chunks or a smaller halo / exchange interval")
    return chunks


def resolve_schedule(iterations, n_fft, hop, halo_frames=None, exchange_every=None):
    """-> (W halo frames, M iterations between refreshes (== iterations: never), exact?)."""
    K = int(iterations)
    R = dependency_radius(n_fft, hop)
    if exchange_every is not None and int(exchange_every) < 0:
        raise ValueError("exchange_every must be >= 0 (0 = never exchange)")
    if halo_frames is None:
        M = min(10, K) if exchange_every is None else (int(exchange_every) or K)
        M = max(1, min(M, max(K, 1)))
        return M * R, M, True
    W = int(halo_frames)
    if exchange_every is None:
        M = W // R if (R > 0 and W >= R) else K        # widest interval the halo keeps exact, else never
    else:
        M = int(exchange_every) or K
    M = max(1, min(M, max(K, 1)))
    return W, M, W >= M * R


class LongformDriver:
    """Device-agnostic bookkeeping of the chunked loop: local signals, halo refreshes, stitching.

    step(chunk, x_local (1, L_local), n_iters) -> (1, L_local) runs the Griffin-Lim iterations on
    one chunk (product: Plan.run on the chunk's magnitude rows).  stitch_add(out, piece, offset,
    fade_in, fade_out) accumulates a weighted piece (product: gl_stitch_add).  `mine` lists the
    chunk indices this process holds: [rank] under torch.distributed, all of them otherwise."""

    def __init__(self, n_fft, hop, n_frames, chunks, mine, step, stitch_add, group=None):
        self.N, self.H, self.T = int(n_fft), int(hop), int(n_frames)
        self.L = self.T * self.H + self.N
        self.chunks = list(chunks)
        self.mine = list(mine)
        self.step = step
        self.stitch_add = stitch_add
        self.group = group
        self.x = {}

    # -- geometry (global sample coordinates) ---------------------------------------------
    def local_range(self, ch):
        return ch.la * self.H, ch.lb * self.H + self.N

    def owned_range(self, ch):
        last = ch.index == len(self.chunks) - 1
        return ch.a * self.H, (self.L if last else ch.b * self.H)

    def _halo_left(self, ch):      # samples chunk ch reads but its left neighbour owns
        return ch.la * self.H, ch.a * self.H

    def _halo_right(self, ch):     # ... its right neighbour owns (up to the last sample a local frame reads)
        if ch.index == len(self.chunks) - 1:
            return ch.b * self.H, ch.b * self.H
        return ch.b * self.H, ch.lb * self.H + self.N - self.H

    def _view(self, ch, lo, hi):
        s0 = self.local_range(ch)[0]
        return self.x[ch.index][0, lo - s0:hi - s0]

    # -- loop --------------------------------------------------------------------------------
    def set_signal(self, ch, x_local):
        lo, hi = self.local_range(ch)
        if tuple(x_local.shape) != (1, hi - lo):
            raise ValueError(f"chunk {ch.index}: start signal must have shape (1, {hi - lo})")
        self.x[ch.index] = x_local

    def run(self, iterations, exchange_every):
        K, M = int(iterations), int(exchange_every)
        done = 0
        while done < K:
            n = min(M, K - done)
            for c in self.mine:
                self.x[c] = self.step(self.chunks[c], self.x[c], n)
            done += n
            if done < K:
                self.refresh()

    def refresh(self):
        """Overwrite every halo with the owning neighbour's samples."""
        C = len(self.chunks)
        if C == 1:
            return
        if len(self.mine) == C:                      # all chunks in this process: plain copies
            for ch in self.chunks:
                if ch.index > 0:
                    lo, hi = self._halo_left(ch)
                    if hi > lo:
                        self._view(ch, lo, hi).copy_(self._view(self.chunks[ch.index - 1], lo, hi))
                if ch.index < C - 1:
                    lo, hi = self._halo_right(ch)
                    if hi > lo:
                        self._view(ch, lo, hi).copy_(self._view(self.chunks[ch.index + 1], lo, hi))
            return
        (r,) = self.mine
        me = self.chunks[r]
        stage = dist.get_backend(self.group) != "nccl" and self.x[r].is_cuda
        ops, landings = [], []

        def send(view, peer):
            buf = view.contiguous()
            ops.append(dist.P2POp(dist.isend, buf.cpu() if stage else buf, self._global_rank(peer), self.group))

        def recv(view, peer):
            buf = torch.empty(view.shape, dtype=view.dtype, device="cpu" if stage else view.device)
            ops.append(dist.P2POp(dist.irecv, buf, self._global_rank(peer), self.group))
            landings.append((view, buf))

        # fixed order on both sides of a boundary: the lower rank's send pairs with the upper's recv
        if r > 0:
            lo, hi = self._halo_right(self.chunks[r - 1])      # what my left neighbour needs from me
            if hi > lo:
                send(self._view(me, lo, hi), r - 1)
            lo, hi = self._halo_left(me)
            if hi > lo:
                recv(self._view(me, lo, hi), r - 1)
        if r < C - 1:
            lo, hi = self._halo_left(self.chunks[r + 1])
            if hi > lo:
                send(self._view(me, lo, hi), r + 1)
            lo, hi = self._halo_right(me)
            if hi > lo:
                recv(self._view(me, lo, hi), r + 1)
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for view, buf in landings:
            view.copy_(buf)

    def _global_rank(self, group_rank):
        return group_rank if self.group is None else dist.get_global_rank(self.group, group_rank)

    # -- output ------------------------------------------------------------------------------
    def piece(self, ch, crossfade):
        """(view of the local signal, global offset, fade_in, fade_out) this chunk contributes."""
        F = int(crossfade)
        lo, hi = self.owned_range(ch)
        first, last = ch.index == 0, ch.index == len(self.chunks) - 1
        plo = lo if first else lo - F // 2
        phi = hi if last else hi + (F - F // 2)
        llo, lhi = self.local_range(ch)
        if plo < llo or phi > lhi - (0 if last else self.H):
            raise ValueError(f"crossfade of {F} samples does not fit inside the halo of chunk {ch.index}")
        return self._view(ch, plo, phi), plo, (0 if first else F), (0 if last else F)

    def assemble(self, crossfade, out):
        """Sum every chunk's weighted piece into the zero-filled `out` (L,); under
        torch.distributed the pieces are all-gathered first so every rank holds the result."""
        C = len(self.chunks)
        if len(self.mine) == C:
            for ch in self.chunks:
                v, off, fi, fo = self.piece(ch, crossfade)
                self.stitch_add(out, v.contiguous(), off, fi, fo)
            return out
        (r,) = self.mine
        meta = []
        for ch in self.chunks:                          # geometry only: every rank can compute it
            lo, hi = self.owned_range(ch)
            F = int(crossfade)
            plo = lo if ch.index == 0 else lo - F // 2
            phi = hi if ch.index == C - 1 else hi + (F - F // 2)
            meta.append((plo, phi - plo, 0 if ch.index == 0 else F, 0 if ch.index == C - 1 else F))
        v, off, fi, fo = self.piece(self.chunks[r], crossfade)
        longest = max(m[1] for m in meta)
        stage = dist.get_backend(self.group) != "nccl" and v.is_cuda
        pad = torch.zeros((longest,), dtype=v.dtype, device="cpu" if stage else v.device)
        pad[:v.numel()] = v
        parts = [torch.empty_like(pad) for _ in range(C)]
        dist.all_gather(parts, pad, group=self.group)
        for (plo, n, pfi, pfo), part in zip(meta, parts):
            self.stitch_add(out, part[:n].to(out.device).contiguous(), plo, pfi, pfo)
        return out


def reconstruct_signal_griffin_lim_longform(magnitude_spectrogram, fft_size, hopsamp, iterations, *, n_chunks=None,
                                            halo_frames=None, exchange_every=None, crossfade=0, x_init=None,
                                            seed=None, rng="numpy", device=None, group=None, gather=True,
                                            run_frames=None):
    """Griffin-Lim on one long utterance, frame-chunked (see the module docstring).

    magnitude_spectrogram: (T, F) numpy / torch, rows = time slices -- the whole utterance on every
    rank (each rank uploads only its rows) -- or a callable rows(la, lb) -> (lb - la, F) fp32
    device tensor passed together with the frame count as ``(rows, T)``.
    n_chunks: world size under torch.distributed (one chunk per rank); otherwise the chunks run in
    turn on the current GPU (default 1 = the plain single-piece loop).
    halo_frames / exchange_every: see resolve_schedule(); the defaults (refresh every 10
    iterations, halo 10*R frames) reproduce the single-piece result up to the fp32 rounding of a
    different summation order at chunk edges.  crossfade: samples over which neighbouring pieces
    are blended at a seam (approximate mode; must fit inside the halo).
    x_init (L,) / seed / rng as in reconstruct_signal_griffin_lim; rng='device' fills each chunk's
    slice of ONE Philox stream (gl_randn_at), so the start signal does not depend on the split.
    run_frames: pin the fused kernels' frames-per-wave-run (gl_plan_set_run_frames) and start every
    chunk on a multiple of it: the chunked result is then bit-identical to a single-piece run with
    the same pin (without it the two differ by fp32 rounding at run seams, which 100 iterations of
    Griffin-Lim amplify to about 1e-3 of the peak).  At n_fft 1024 use an even run length and
    halo_frames = exchange_every * (R + 1): that kernel transforms two frames per complex FFT, so a
    stale frame perturbs its partner's last bits and rounding travels one frame further.
    Returns the (L,) waveform (numpy float64 for numpy input, else an fp32 device tensor); with
    gather=False under torch.distributed: (piece, global_offset, fade_in, fade_out) of this rank."""
    from . import audio_utilities as au
    from .plan import get_plan

    N, H, K = int(fft_size), int(hopsamp), int(iterations)
    F = N // 2 + 1
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    if isinstance(magnitude_spectrogram, tuple) and callable(magnitude_spectrogram[0]):
        rows, T = magnitude_spectrogram
        T = int(T)
        was_np = False
        dev = au._pick_device(device, x_init)
    else:
        dev = au._pick_device(device, magnitude_spectrogram, x_init)
        was_np = not isinstance(magnitude_spectrogram, torch.Tensor)
        full = magnitude_spectrogram
        if full.ndim != 2 or full.shape[1] != F:
            raise ValueError(f"magnitude_spectrogram must be (T, {F})")
        T = int(full.shape[0])

        def rows(la, lb):
            return au._to_device(full[la:lb], dev, torch.float32)[0]
    if T < 1:
        raise ValueError("need at least one frame")
    L = T * H + N
    if distributed:
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        if n_chunks not in (None, world):
            raise ValueError(f"under torch.distributed n_chunks is the world size ({world})")
        C, mine = world, [rank]
    else:
        C = 1 if n_chunks is None else int(n_chunks)
        mine = list(range(C))
    W, M, _exact = resolve_schedule(K, N, H, halo_frames, exchange_every)
    if C == 1:
        W, M = 0, max(K, 1)
    chunks = plan_chunks(T, C, W, N, H, align=run_frames or 1)

    # one shared plan sized for the longest chunk; the run-length pin travels with each call (never plan state)
    longest = max(chunks[c].lb - chunks[c].la for c in mine)
    plan = get_plan(N, H, longest, 1, dev)
    mags = {}
    for c in mine:
        ch = chunks[c]
        mags[c] = rows(ch.la, ch.lb).reshape(1, ch.lb - ch.la, F).contiguous()

    def step(ch, x_local, n):
        return plan.run(mags[ch.index], x_local, n, out=x_local, run_frames=run_frames or 0)

    def stitch(out, piece, offset, fade_in, fade_out):
        plan.stitch_add(out, piece, offset, fade_in, fade_out)

    drv = LongformDriver(N, H, T, chunks, mine, step, stitch, group)

    # start signal: one global draw, every chunk takes its slice (audio_utilities.py:176)
    host0 = None
    if x_init is not None:
        if tuple(x_init.shape) != (L,):
            raise ValueError(f"x_init must have shape ({L},)")
    elif rng == "numpy":
        if seed is not None:
            host0 = np.random.RandomState(seed).randn(L).astype(np.float32)
        else:
            t = torch.empty((L,), dtype=torch.float32)
            if not distributed or dist.get_rank(group) == 0:
                t = torch.from_numpy(np.random.randn(L).astype(np.float32))
            if distributed:                      # ranks do not share numpy's global RNG state
                stage = dist.get_backend(group) == "nccl"
                tt = t.to(dev) if stage else t
                dist.broadcast(tt, src=drv._global_rank(0), group=group)
                t = tt.cpu() if stage else tt
            host0 = t.numpy()
    elif rng != "device":
        raise ValueError("rng must be 'numpy' or 'device'")
    for c in mine:
        ch = chunks[c]
        lo, hi = drv.local_range(ch)
        if x_init is not None:
            x0 = au._to_device(x_init[lo:hi], dev, torch.float32)[0].clone()
        elif host0 is not None:
            x0 = torch.from_numpy(host0[lo:hi]).to(dev)
        else:
            x0 = plan.randn((hi - lo,), 0 if seed is None else seed, offset=lo)
        drv.set_signal(ch, x0.reshape(1, hi - lo))

    drv.run(K, M)

    if distributed and not gather:
        v, off, fi, fo = drv.piece(chunks[mine[0]], crossfade)
        return v, off, fi, fo
    if C == 1:
        out = drv.x[0][0]
    else:
        out = torch.zeros((L,), dtype=torch.float32, device=dev)
        drv.assemble(crossfade, out)
    if was_np:
        return out.cpu().numpy().astype(np.float64)
    return out
