import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from tacotron2_amd.plan import Plan
N, H, B = 2048, 256, 64
for T in (215, 431, 861, 1722):
    plan = Plan(N, H, T, B)
    rng = np.random.default_rng(1)
    mag = torch.from_numpy(np.abs(rng.standard_normal((B, T, N // 2 + 1))).astype(np.float32)).cuda()
    x0 = plan.randn((B, plan.signal_length(T)), seed=2)
    x = torch.empty_like(x0)
    K = 60
    for _ in range(2): plan.run(mag, x0, K, out=x)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): plan.run(mag, x0, K, out=x)
    e1.record(); torch.cuda.synchronize()
    print("T", T, "us per iteration", e0.elapsed_time(e1) / 5 / K * 1e3)
    del plan
