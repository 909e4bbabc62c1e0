import sys, torch
sys.path.insert(0, "/root/repo")
from tacotron2_amd import filters
from tacotron2_amd.plan import Plan
from bench import synth_mels
dev = torch.device("cuda:0")
B, T, N, H, K = 32, 1001, 1024, 275, 12
plan = Plan(N, H, T, B, dev)
fb = torch.from_numpy(filters.mel(22050, N, 80, 125, 7600)).to(dev)
mel = torch.from_numpy(synth_mels(B, T, 1)).to(dev)
mag = plan.mel_to_mag(fb, mel)
x0 = plan.randn((B, T * H + N), 3)
x = plan.run(mag, x0, K)
torch.cuda.synchronize()
print("finite", bool(torch.isfinite(x).all()))
