import sys, torch
sys.path.insert(0, "/root/repo")
from tacotron2_amd import filters
from tacotron2_amd.plan import Plan
dev = torch.device("cuda:0")
B, N, n = 64, 2048, 220500
plan = Plan(N, N // 4, 1 + n // (N // 4), B, dev)
fb = torch.from_numpy(filters.mel(22050, N, 80, 125, 7600)).to(dev)
y = torch.randn((B, n), device=dev) * 0.1
for _ in range(3): plan.wav_to_logmel(y, fb)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): plan.wav_to_logmel(y, fb)
e1.record(); torch.cuda.synchronize()
print("wav_to_logmel us", e0.elapsed_time(e1) / 10 * 1e3, "audio-s/s", B * n / 22050 / (e0.elapsed_time(e1) / 10 * 1e-3))
