import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from tacotron2_amd.plan import Plan
N, H = 1024, 275
for (T, B, G) in [(12, 1, 9), (40, 2, 9), (40, 2, 16)]:
    plan = Plan(N, H, T, B)
    rng = np.random.default_rng(1)
    mag = torch.from_numpy(np.abs(rng.standard_normal((B, T, N // 2 + 1))).astype(np.float32)).cuda()
    x0 = plan.randn((B, plan.signal_length(T)), seed=2)
    ref = plan.run(mag, x0, 1, path=1)
    got = plan.run(mag, x0, 1, run_frames=G)
    err = (got - ref).abs()[0].cpu().numpy()
    bad = np.nonzero(err > 1e-4 * float(ref.abs().max()))[0]
    print("T", T, "G", G, "max err", err.max(), "n bad", bad.size, "first/last bad", (bad[:1], bad[-1:]) if bad.size else None)
    if bad.size:
        # contiguous ranges
        cuts = np.nonzero(np.diff(bad) > 1)[0]
        starts = np.r_[bad[0], bad[cuts + 1]]; ends = np.r_[bad[cuts], bad[-1]]
        print("  ranges", list(zip(starts.tolist(), ends.tolist()))[:12])
