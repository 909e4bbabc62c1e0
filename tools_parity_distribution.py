#!/usr/bin/env python3
"""How often, and by how much, does the fp32 HIP path miss the north-star tolerance on BASELINE's own workload?

    python tools_parity_distribution.py [--per-shard 9] [--workers 16] [--out gpurun_out/r3_parity_distribution.json]

Takes utterances of the config-3 batch of 1024 (global ids spread evenly over all 8 shards of the 8-GPU split,
always including 389 -- the one utterance known to sit at the edge), T = 861, n_fft 2048, hop 256, 60 iterations,
and compares, per utterance, against the float64 oracle (audio_utilities.py:154-188 restated, oracle/gl_oracle.py)
on the SAME magnitudes and start signal:
  * the GPU result with the run length pinned (27 frames, what a sharded job pins) and automatic, and of the
    any-N kernels (an independent second implementation); with --lib TAG=PATH also of other builds of the library,
  * numpy's own float32 run of the reference algorithm (griffin_lim_vec_f32: pocketfft in single precision),
the yardstick for what float32 arithmetic can hold.  Prints one line per utterance and a summary; the JSON record
is what DESIGN.md section 2 and the bound in tests/test_gpu_parity.py (assert_gl_parity) are derived from.

TEST INFRASTRUCTURE: imports oracle/ as the checker.  The GPU phase runs in one child process per library and the
CPU legs in child processes too (one utterance per task, --workers at a time), all started with subprocess by a
parent that never touches the GPU.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import numpy as np  # noqa: E402

N, H, T, K = 2048, 256, 861, 60
L = T * H + N
TOL = 1e-3


def pick_ids(per_shard, shards=8, total=1024):
    """per_shard ids from each of the `shards` contiguous shards of the 1024-batch, evenly spaced; 389 always in."""
    size = total // shards
    ids = []
    for s in range(shards):
        ids += [s * size + int(round((j + 0.5) * size / per_shard)) for j in range(per_shard)]
    ids = sorted(set(ids) | {389})
    return ids


def worker(tmpdir, i):
    """CPU leg for utterance row i: float64 oracle, float32 numpy run, errors of every GPU result in tmpdir."""
    import gl_oracle as orc
    ids = np.load(os.path.join(tmpdir, "ids.npy"))
    mag32 = np.load(os.path.join(tmpdir, "mag.npy"), mmap_mode="r")[i]
    x032 = np.load(os.path.join(tmpdir, "x0.npy"), mmap_mode="r")[i]
    m64, x64 = mag32.astype(np.float64), x032.astype(np.float64)
    ref = orc.griffin_lim_vec(m64, N, H, K, x64)
    peak = float(np.max(np.abs(ref)))
    y32 = orc.griffin_lim_vec_f32(np.array(mag32), N, H, K, np.array(x032)).astype(np.float64)
    sc_ref = orc.spectral_convergence(ref, m64, N, H)
    out = {"id": int(ids[i]), "peak": peak}
    e = np.abs(y32 - ref) / peak
    out["f32"] = {"err": float(e.max()), "at": int(e.argmax())}
    for f in sorted(os.listdir(tmpdir)):
        if not f.startswith("y_"):
            continue
        got = np.load(os.path.join(tmpdir, f), mmap_mode="r")[i].astype(np.float64)
        e = np.abs(got - ref) / peak
        out[f[2:-4]] = {"err": float(e.max()), "at": int(e.argmax()), "p999": float(np.quantile(e, 0.999)),
                        "sc_delta": float(abs(orc.spectral_convergence(got, m64, N, H) - sc_ref))}
    print(json.dumps(out), flush=True)


VARIANTS = {"pin27": {"run_frames": 27}, "auto": {}}


def gpu_child(tmpdir, tag):
    """One process per library (GL_HIP_LIB selects it at import): the Griffin-Lim results of every variant for the
    ids in tmpdir; the first child also writes the magnitudes (GPU mel inversion, the measured path's own) and
    the start signals."""
    import torch
    from bench import synth_mels_by_id
    from tacotron2_amd import filters
    from tacotron2_amd.plan import Plan
    ids = [int(g) for g in np.load(os.path.join(tmpdir, "ids.npy"))]
    B = len(ids)
    dev = torch.device("cuda", 0)
    fb = torch.from_numpy(filters.mel(22050, N, 80, 125, 7600)).to(dev)
    plan = Plan(N, H, T, B, dev)
    mel = torch.from_numpy(synth_mels_by_id(ids, T)).to(dev)
    x0 = torch.cat([plan.randn((1, L), seed=42, offset=g * L) for g in ids])
    mag = plan.mel_to_mag(fb, mel)
    variants = dict(VARIANTS)
    if tag == "gpu":
        variants["generic"] = {"path": 1}             # the any-N kernels: an independent second implementation
    for name, kw in variants.items():
        np.save(os.path.join(tmpdir, f"y_{tag}_{name}.npy"), plan.run(mag, x0, K, **kw).cpu().numpy())
    if not os.path.exists(os.path.join(tmpdir, "mag.npy")):
        np.save(os.path.join(tmpdir, "mag.npy"), mag.cpu().numpy())
        np.save(os.path.join(tmpdir, "x0.npy"), x0.cpu().numpy())
    plan.close()


def gpu_phase(ids, tmpdir, libs):
    np.save(os.path.join(tmpdir, "ids.npy"), np.asarray(ids, dtype=np.int64))
    for tag, lib in libs.items():
        env = dict(os.environ)
        if lib:
            env["GL_HIP_LIB"] = lib
        pr = subprocess.run([sys.executable, os.path.abspath(__file__), "--gpu-child", tmpdir, tag], env=env, timeout=300)
        if pr.returncode != 0:
            raise RuntimeError(f"GPU phase of library {tag} failed with exit code {pr.returncode}")


def cpu_phase(tmpdir, n, workers):
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
    rows, running, todo = [], [], list(range(n))
    while todo or running:
        while todo and len(running) < workers:
            i = todo.pop(0)
            running.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", tmpdir, str(i)],
                                            stdout=subprocess.PIPE, env=env, text=True))
        pr = running.pop(0)
        out, _ = pr.communicate(timeout=900)
        if pr.returncode != 0:
            raise RuntimeError(f"parity worker failed with exit code {pr.returncode}")
        rows.append(json.loads(out.strip().splitlines()[-1]))
        print("  ".join([f"id {rows[-1]['id']:4d}", f"f32 {rows[-1]['f32']['err']:.2e}"] +
                        [f"{k} {v['err']:.2e}" for k, v in rows[-1].items() if isinstance(v, dict) and k != "f32"]),
              file=sys.stderr, flush=True)
    return sorted(rows, key=lambda r: r["id"])


def summarise(rows, names):
    s = {"n": len(rows), "tolerance": TOL, "geometry": {"n_fft": N, "hop": H, "frames": T, "iters": K}}
    e32 = np.array([r["f32"]["err"] for r in rows])
    s["numpy_f32"] = {"over_tol": int((e32 > TOL).sum()), "worst": float(e32.max()), "median": float(np.median(e32))}
    for nm in names:
        e = np.array([r[nm]["err"] for r in rows])
        ratio = e / e32
        s[nm] = {"over_tol": int((e > TOL).sum()), "over_tol_ids": [r["id"] for r in rows if r[nm]["err"] > TOL],
                 "worst": float(e.max()), "worst_id": int(rows[int(e.argmax())]["id"]), "median": float(np.median(e)),
                 "p90": float(np.quantile(e, 0.9)),
                 "ratio_to_numpy_f32": {"median": float(np.median(ratio)), "geomean": float(np.exp(np.mean(np.log(ratio)))),
                                        "max": float(ratio.max()), "min": float(ratio.min()),
                                        "max_where_err_over_5e-4": float(max([q for q, v in zip(ratio, e) if v > 5e-4] or [0.0]))},
                 "worst_sc_delta": float(max(r[nm]["sc_delta"] for r in rows))}
    return s


def run(ids, workers, libs=None):
    """libs: {tag: path of a libgl_hip build, or None for the product library}; results are named <tag>_<variant>."""
    libs = libs or {"gpu": None}
    with tempfile.TemporaryDirectory() as tmp:
        t0 = time.perf_counter()
        gpu_phase(ids, tmp, libs)
        t1 = time.perf_counter()
        rows = cpu_phase(tmp, len(ids), workers)
        t2 = time.perf_counter()
    names = [k for k in rows[0] if isinstance(rows[0][k], dict) and k != "f32"]
    return {"summary": summarise(rows, names), "ids": ids, "utterances": rows,
            "seconds": {"gpu_phase": t1 - t0, "cpu_phase": t2 - t1, "workers": workers}}


def main():
    if len(sys.argv) == 4 and sys.argv[1] == "--worker":
        return worker(sys.argv[2], int(sys.argv[3]))
    if len(sys.argv) == 4 and sys.argv[1] == "--gpu-child":
        return gpu_child(sys.argv[2], sys.argv[3])
    ap = argparse.ArgumentParser()
    ap.add_argument("--per-shard", type=int, default=16, help="utterances from each of the 8 shards (16 -> 128 + id 389)")
    ap.add_argument("--workers", type=int, default=16)
    ap.add_argument("--lib", action="append", default=[], metavar="TAG=PATH",
                    help="also run this build of the library (attribution studies, abl/build.sh)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r3_parity_distribution.json"))
    args = ap.parse_args()
    ids = pick_ids(args.per_shard)
    libs = {"gpu": None}
    for spec in args.lib:
        tag, path = spec.split("=", 1)
        libs[tag] = os.path.abspath(path)
    rec = run(ids, args.workers, libs)
    from bench import file_sha256
    lib_path = os.environ.get("GL_HIP_LIB") or os.path.join(ROOT, "tacotron2_amd", "libgl_hip.so")
    rec["lib_sha256"] = file_sha256(lib_path)
    rec["libs"] = {k: (v or lib_path) for k, v in libs.items()}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as fh:
        json.dump(rec, fh, indent=1)
    print(json.dumps(rec["summary"], indent=1))


if __name__ == "__main__":
    main()
