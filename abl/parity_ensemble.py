"""How wide is the float32 error distribution of ONE utterance?  For each id: the float64 oracle, M numpy-float32
runs of the reference algorithm whose start signals differ by one float32 rounding (x0 * (1 + 6e-8 g), g ~ N(0,1):
a different rounding trajectory, the same problem), and the GPU results (pinned / automatic run length, any-N
kernels).  Also the growth of the error with the iteration count (20 / 40 / 60).  Study behind assert_gl_parity.
usage: python abl/parity_ensemble.py <M> id id ...   (GPU phase in a child process, CPU legs in worker processes)"""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
N, H, T = 2048, 256, 861
L = T * H + N
KS = (20, 40, 60)

def gpu_child(tmp, ids):
    import torch
    from bench import synth_mels_by_id
    from tacotron2_amd import filters
    from tacotron2_amd.plan import Plan
    fb = torch.from_numpy(filters.mel(22050, N, 80, 125, 7600)).cuda()
    p = Plan(N, H, T, len(ids))
    mel = torch.from_numpy(synth_mels_by_id(ids, T)).cuda()
    x0 = torch.cat([p.randn((1, L), seed=42, offset=g * L) for g in ids])
    mag = p.mel_to_mag(fb, mel)
    for K in KS:
        for name, kw in (("pin27", {"run_frames": 27}), ("auto", {}), ("generic", {"path": 1})):
            np.save(os.path.join(tmp, f"y_{name}_{K}.npy"), p.run(mag, x0, K, **kw).cpu().numpy())
    np.save(os.path.join(tmp, "mag.npy"), mag.cpu().numpy()); np.save(os.path.join(tmp, "x0.npy"), x0.cpu().numpy())

def worker(tmp, i, m, M):
    import gl_oracle as orc
    mag = np.array(np.load(os.path.join(tmp, "mag.npy"), mmap_mode="r")[i]); x0 = np.array(np.load(os.path.join(tmp, "x0.npy"), mmap_mode="r")[i])
    out = {"i": i, "m": m}
    if m < 0:      # float64 references at every K + GPU errors
        x = x0.astype(np.float64); refs = {}
        for K in KS:
            x = orc.griffin_lim_vec(mag.astype(np.float64), N, H, K - (KS[KS.index(K) - 1] if KS.index(K) else 0), x)
            refs[K] = x
            np.save(os.path.join(tmp, f"ref_{i}_{K}.npy"), x)
        for K in KS:
            pk = np.max(np.abs(refs[K]))
            for name in ("pin27", "auto", "generic"):
                y = np.load(os.path.join(tmp, f"y_{name}_{K}.npy"), mmap_mode="r")[i].astype(np.float64)
                out[f"{name}_{K}"] = float(np.max(np.abs(y - refs[K])) / pk)
    else:
        g = np.random.default_rng([i, m]).standard_normal(L)
        xp = (x0.astype(np.float64) * (1.0 + (6e-8 * g if m else 0.0))).astype(np.float32)
        import scipy.fft as sf
        # float32 run with snapshots
        w = np.hanning(N).astype(np.float32); x = xp.copy()
        idx = (np.arange(T) * H)[:, None] + np.arange(N)[None, :]
        for it in range(1, KS[-1] + 1):
            S = sf.rfft(w[None, :] * x[idx], axis=1); a = np.abs(S); z = a == 0; a[z] = 1
            P = (S / a) * mag; P[z] = mag[z]
            fr = sf.irfft(P, n=N, axis=1).astype(np.float32) * w[None, :]
            x = np.zeros(L, np.float32)
            for n in range(T): x[n * H:n * H + N] += fr[n]
            if it in KS: out[f"snap_{it}"] = x.copy()
    return out

def main():
    if sys.argv[1] == "--gpu":
        return gpu_child(sys.argv[2], [int(a) for a in sys.argv[3:]])
    if sys.argv[1] == "--worker":
        tmp, i, m, M = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
        out = worker(tmp, i, m, M)
        snaps = {k: out.pop(k) for k in list(out) if k.startswith("snap_")}
        for k, v in snaps.items(): np.save(os.path.join(tmp, f"f32_{i}_{m}_{k[5:]}.npy"), v)
        print(json.dumps(out)); return
    M = int(sys.argv[1]); ids = [int(a) for a in sys.argv[2:]]
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.run([sys.executable, __file__, "--gpu", tmp] + [str(a) for a in ids], check=True, timeout=300)
        tasks = [(i, m) for i in range(len(ids)) for m in range(-1, M)]
        running, res = [], {}
        env = dict(os.environ, OMP_NUM_THREADS="1")
        while tasks or running:
            while tasks and len(running) < 16:
                i, m = tasks.pop(0)
                running.append(subprocess.Popen([sys.executable, __file__, "--worker", tmp, str(i), str(m), str(M)], stdout=subprocess.PIPE, text=True, env=env))
            pr = running.pop(0); o, _ = pr.communicate(timeout=900); assert pr.returncode == 0
            d = json.loads(o.strip().splitlines()[-1]); res[(d["i"], d["m"])] = d
        for i, g in enumerate(ids):
            line = {"id": g}
            for K in KS:
                ref = np.load(os.path.join(tmp, f"ref_{i}_{K}.npy")); pk = np.max(np.abs(ref))
                es = [float(np.max(np.abs(np.load(os.path.join(tmp, f"f32_{i}_{m}_{K}.npy")).astype(np.float64) - ref)) / pk) for m in range(M)]
                line[f"f32_{K}"] = [float("%.2e" % e) for e in es]
                for name in ("pin27", "auto", "generic"):
                    line[f"{name}_{K}"] = float("%.2e" % res[(i, -1)][f"{name}_{K}"])
            print(json.dumps(line), flush=True)

if __name__ == "__main__":
    main()
