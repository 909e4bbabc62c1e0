"""Helper run as a subprocess by test_gpu_parity.py with GL_P1_VERSION=1|2|f: n_fft-1024 fused kernel (chosen
version forced; f = the per-pair frames form, which needs an unpinned run length) against the float64 oracle on
small cases, including an all-zero frame next to a live one, odd frame counts and runs with tails."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gl_oracle as orc  # noqa: E402  (checker only)
from tacotron2_amd.plan import Plan  # noqa: E402

N = 1024
worst = 0.0
for (H, B, T, K, G) in [(275, 2, 41, 6, 0), (275, 1, 7, 3, 4), (256, 3, 64, 5, 6), (100, 1, 30, 4, 11), (275, 2, 1, 2, 0),
                        # every reach floor(1023/H) = 1..8 of the per-pair form (hop 100 has 10: run kernels)
                        (512, 2, 9, 3, 0), (400, 1, 12, 3, 0), (220, 1, 19, 3, 0), (200, 1, 33, 4, 0), (170, 2, 14, 3, 0), (146, 1, 25, 3, 0),
                        (128, 1, 40, 3, 0), (114, 2, 21, 2, 0), (1024, 1, 5, 2, 0)]:
    rng = np.random.default_rng(T + H)
    mag = np.abs(rng.standard_normal((B, T, N // 2 + 1))).astype(np.float32)
    x0 = rng.standard_normal((B, T * H + N)).astype(np.float32)
    if T > 8:
        x0[0, 3 * H:3 * H + N] = 0.0          # frame 3 of utterance 0 is exactly zero at the first iteration
    p = Plan(N, H, T, B, "cuda:0")
    p.set_run_frames(0 if os.environ.get("GL_P1_VERSION") == "f" else G)
    x = p.run(torch.from_numpy(mag).cuda(), torch.from_numpy(x0).cuda(), K).cpu().numpy()
    for b in range(B):
        ref = orc.griffin_lim_vec(mag[b].astype(np.float64), N, H, K, x0[b].astype(np.float64))
        worst = max(worst, float(np.max(np.abs(x[b] - ref)) / np.max(np.abs(ref))))
    p.close()
print("worst", worst)
sys.exit(0 if worst <= 1e-3 else 1)
