"""N>1 path on CPU: world_size-2 gloo process group, per-utterance sharding + final gather.
The reconstruction itself is replaced by a deterministic stand-in (no GPU here); what is under
test is the partition, the ragged gather and that sharded == unsharded bitwise."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tacotron2_amd import sharding


def _fake_reconstruct(mels):
    # deterministic per-utterance function of the input only (like Griffin-Lim with a fixed x0)
    t = torch.as_tensor(mels, dtype=torch.float32)
    return torch.cumsum(t.reshape(t.shape[0], -1), dim=1) * 0.5 + 1.0


def _worker(rank, world, port, n_items, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(5)
        mels = rng.standard_normal((n_items, 4, 6)).astype(np.float32)     # same on every rank
        full = sharding.reconstruct_sharded(mels, _fake_reconstruct, gather=True)
        lo, hi = sharding.shard_bounds(n_items, world, rank)
        q.put((rank, lo, hi, full.numpy()))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_shard_bounds_partition():
    for n in (0, 1, 2, 7, 64, 1024, 1025):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_bounds(4, 2, 2)


@pytest.mark.parametrize("n_items", [5, 8])
def test_world_size_2_sharded_equals_unsharded(n_items):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(5)
    mels = rng.standard_normal((n_items, 4, 6)).astype(np.float32)
    ref = _fake_reconstruct(mels).numpy()
    spans = sorted((lo, hi) for _, lo, hi, _ in results)
    assert spans[0][0] == 0 and spans[-1][1] == n_items and spans[0][1] == spans[1][0]
    for _, _, _, full in results:
        assert full.shape == ref.shape
        assert np.array_equal(full, ref)          # bitwise: sharding changes nothing


def _subgroup_worker(rank, world, port, n_items, members, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        grp = dist.new_group(members)             # every rank takes part in creating it
        if rank in members:
            rng = np.random.default_rng(5)
            mels = rng.standard_normal((n_items, 4, 6)).astype(np.float32)
            full = sharding.reconstruct_sharded(mels, _fake_reconstruct, gather=True, group=grp)
            q.put((rank, full.numpy()))
    finally:
        dist.destroy_process_group()


def test_sub_group_shards_and_gathers_by_group_rank():
    """A job that gives Griffin-Lim a sub-group of the world (ranks 2 and 0 of 3, in that order): shard bounds and
    the gather both follow the GROUP's size and ranks."""
    world, members, n_items = 3, [2, 0], 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_subgroup_worker, args=(r, world, port, n_items, members, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in members]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(5)
    ref = _fake_reconstruct(rng.standard_normal((n_items, 4, 6)).astype(np.float32)).numpy()
    assert sorted(r for r, _ in results) == [0, 2]
    for _, full in results:
        assert np.array_equal(full, ref)
