import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import numpy as np, torch
import gl_oracle as orc
from tacotron2_amd.plan import Plan
N, H, K = 2048, 256, 8
for (B, T) in [(3, 61), (1, 7), (2, 200), (1, 1), (2, 8)]:
    rng = np.random.default_rng(T)
    mag = np.abs(rng.standard_normal((B, T, N // 2 + 1))).astype(np.float32)
    x0 = rng.standard_normal((B, T * H + N)).astype(np.float32)
    p = Plan(N, H, T, B, "cuda:0")
    x = p.run(torch.from_numpy(mag).cuda(), torch.from_numpy(x0).cuda(), K).cpu().numpy()
    err = 0
    for b in range(B):
        ref = orc.griffin_lim_vec(mag[b].astype(np.float64), N, H, K, x0[b].astype(np.float64))
        err = max(err, np.max(np.abs(x[b] - ref)) / np.max(np.abs(ref)))
    print("B", B, "T", T, "max rel err", err)
    p.close()
