"""RCCL on the real stack with the one GPU a test box has: a world of ONE rank over the "nccl" backend
(= RCCL on ROCm).  Proves process-group init on the device, a barrier, an all_reduce, and the padded
all_gather of sharding.gather_waveforms on device tensors that the library's kernels have just written on
torch's current stream (stream ordering between libgl_hip's launches and RCCL's).  N > 1 stays unmeasured.
Run as a child process by tests/test_round3.py and by bench.py's BENCH_FORCE_DIST=1 leg; prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        from bench import synth_mels_by_id
        from tacotron2_amd import griffin_lim, sharding
        dist.barrier()
        t = torch.tensor([3.25], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t.item()) == 3.25
        N, H, T, K, B = 2048, 256, 120, 8, 5
        mels = torch.from_numpy(synth_mels_by_id(range(B), T)).to(dev)
        x0 = torch.from_numpy(np.random.default_rng(3).standard_normal((B, T * H + N)).astype(np.float32)).to(dev)

        def reconstruct(local):
            x, _ = griffin_lim.mel_to_audio(local, n_fft=N, hop=H, iterations=K, x_init=x0[:len(local)].clone(),
                                            normalize=None, run_frames=27)
            return x

        local = reconstruct(mels).clone()
        # no synchronisation between the library's launches and the collective: RCCL must order itself
        # behind torch's current stream, which is the stream libgl_hip launched on
        full = sharding.reconstruct_sharded(mels, reconstruct, gather=True, always_collective=True)
        torch.cuda.synchronize()
        ok = bool(torch.equal(full, local)) and tuple(full.shape) == (B, T * H + N) and full.is_cuda
        print(json.dumps({"rccl_world1": ok, "backend": dist.get_backend(), "world": dist.get_world_size(),
                          "shape": list(full.shape), "finite": bool(torch.isfinite(full).all().item())}))
        return 0 if ok else 1
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
