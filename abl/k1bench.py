import sys, time, torch
sys.path.insert(0, "/root/repo")
from tacotron2_amd import filters
from tacotron2_amd.plan import Plan
from bench import synth_mels
dev = torch.device("cuda:0")
B, T, N = 64, 861, 2048
plan = Plan(N, 256, T, B, dev)
fb = torch.from_numpy(filters.mel(22050, N, 80, 125, 7600)).to(dev)
mel = torch.from_numpy(synth_mels(B, T, 1)).to(dev)
mag = torch.empty((B, T, N // 2 + 1), device=dev)
for _ in range(3): plan.mel_to_mag(fb, mel, out=mag)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): plan.mel_to_mag(fb, mel, out=mag)
e1.record(); torch.cuda.synchronize()
print("mel_to_mag us", e0.elapsed_time(e1) / 20 * 1e3)
