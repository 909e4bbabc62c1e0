"""Where does the fp32-vs-float64 error of one config-3 utterance come from?  GPU run vs float64 oracle,
float64 oracle with a 1e-7 perturbed start, and a numpy/pocketfft fp32 restatement, same inputs."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, scipy.fft as sf, torch
import gl_oracle as orc
from bench import synth_mels_by_id
from tacotron2_amd import filters
from tacotron2_amd.plan import Plan

N, H, T, K = 2048, 256, 861, 60
L = T * H + N
ids = [int(a) for a in sys.argv[1:]] or [389, 511]
fb = torch.from_numpy(filters.mel(22050, N, 80, 125, 7600)).cuda()
p = Plan(N, H, T, len(ids))
mel = torch.from_numpy(synth_mels_by_id(ids, T)).cuda()
x0 = torch.cat([p.randn((1, L), seed=42, offset=g * L) for g in ids])
mag = p.mel_to_mag(fb, mel)
w64 = np.hanning(N)

def gl32(mag, x0, K):
    w = w64.astype(np.float32); x = x0.astype(np.float32)
    idx = (np.arange(T) * H)[:, None] + np.arange(N)[None, :]
    for _ in range(K):
        S = sf.rfft(w[None, :] * x[idx], axis=1)
        a = np.abs(S); z = a == 0; a[z] = 1
        P = (S / a) * mag; P[z] = mag[z]
        fr = sf.irfft(P, n=N, axis=1).astype(np.float32) * w[None, :]
        x = np.zeros(L, np.float32)
        for n in range(T): x[n * H:n * H + N] += fr[n]
    return x

for Kc in (20, 40, 60):
    y = {}
    for name, kw in (("auto", {}), ("pin27", {"run_frames": 27}), ("generic", {"path": 1})):
        y[name] = p.run(mag, x0, Kc, **kw).cpu().numpy()
    for i, g in enumerate(ids):
        m64 = mag[i].cpu().numpy().astype(np.float64)
        x64 = x0[i].cpu().numpy().astype(np.float64)
        ref = orc.griffin_lim_vec(m64, N, H, Kc, x64)
        pk = np.max(np.abs(ref))
        rng = np.random.default_rng(g)
        refp = orc.griffin_lim_vec(m64, N, H, Kc, x64 * (1 + 1e-7 * rng.standard_normal(L)))
        y32 = gl32(mag[i].cpu().numpy(), x0[i].cpu().numpy(), Kc)
        line = f"id {g} K={Kc}: f64 perturbed {np.max(np.abs(refp-ref))/pk:.2e}  numpy-fp32 {np.max(np.abs(y32-ref))/pk:.2e}"
        for name in y:
            e = np.abs(y[name][i] - ref) / pk
            line += f"  gpu-{name} {e.max():.2e} (at {int(e.argmax())}, p99.9 {np.quantile(e, 0.999):.1e})"
        print(line, flush=True)
