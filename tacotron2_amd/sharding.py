"""Per-utterance sharding of a batch over the GPUs of one node.

Utterances are independent in Griffin-Lim, so the data path needs no
collective: rank r reconstructs utterances [lo, hi) of the batch on its own GPU.  The only
communication is an optional final gather of the waveforms (RCCL all_gather when the process
group is "nccl", gloo on CPU in the tests).  One process per GPU, launched with
``python -m torch.distributed.run``.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_items, world_size, rank):
    """Contiguous balanced split: the first n_items % world_size ranks get one extra item."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, extra = divmod(int(n_items), int(world_size))
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def shard_batch(batch, world_size=None, rank=None, group=None):
    """Slice dim 0 of `batch` (tensor or ndarray) for this rank (of `group`, default: the world)."""
    if world_size is None:
        world_size = dist.get_world_size(group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds(len(batch), world_size, rank)
    return batch[lo:hi]


def gather_waveforms(local, n_total, group=None, always_collective=False):
    """All-gather per-rank waveforms (n_local, L) into (n_total, L) in batch order.

    Shards may differ by one utterance, so every rank pads to the largest shard; the padding is
    dropped after the gather.  Returns the full tensor on every rank.  A group of one rank returns
    `local` without communicating unless always_collective asks for the all_gather anyway (a one-GPU
    box can then exercise the RCCL path: init, stream ordering against the library's stream, the
    padded gather)."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not always_collective):
        return local
    world = dist.get_world_size(group)
    max_local = (n_total + world - 1) // world
    L = local.shape[1]
    padded = torch.zeros((max_local, L), dtype=local.dtype, device=local.device)
    padded[:local.shape[0]] = local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    out = []
    for r, part in enumerate(parts):
        lo, hi = shard_bounds(n_total, world, r)
        out.append(part[:hi - lo])
    return torch.cat(out, dim=0)


def reconstruct_sharded(mels, reconstruct_fn, gather=True, group=None, always_collective=False):
    """Run `reconstruct_fn(local_mels) -> (n_local, L) tensor` on this rank's shard of `mels`
    (B, n_mels, T) and optionally gather the waveforms of the whole batch."""
    n_total = len(mels)
    local = reconstruct_fn(shard_batch(mels, group=group))
    return gather_waveforms(local, n_total, group, always_collective) if gather else local
