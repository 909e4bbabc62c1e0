import sys, time, torch
sys.path.insert(0, "/root/repo")
from tacotron2_amd import filters
from tacotron2_amd.plan import Plan
from bench import synth_mels
dev = torch.device("cuda:0")
for (B, T, tag) in [(32, 1001, "config 5"), (1, 641, "config 1")]:
    N, H, K = 1024, 275, 60
    plan = Plan(N, H, T, B, dev)
    fb = torch.from_numpy(filters.mel(22050, N, 80, 125, 7600)).to(dev)
    mel = torch.from_numpy(synth_mels(B, T, 1)).to(dev)
    mag = plan.mel_to_mag(fb, mel)
    x0 = plan.randn((B, T * H + N), 3)
    x = torch.empty_like(x0)
    for _ in range(2): plan.run(mag, x0, K, out=x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): plan.run(mag, x0, K, out=x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(tag, "ms per 60 iterations", dt * 1e3, "audio-s/s", B * (T * H + N) / 22050 / dt, "finite", bool(torch.isfinite(x).all()))
    plan.close()
