#!/usr/bin/env python3
"""Headline benchmark: audio-seconds reconstructed per second, Griffin-Lim 60
iterations, n_fft 2048 / hop 256, 80-band synthetic mels (BASELINE.json).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch that is already resident in
HBM: mel (B,80,T) -> linear magnitude on MFMA, 60 fused Griffin-Lim iterations
from a resident start signal, peak normalisation (gl_peak_normalise).  The first
two are ONE library call (gl_mel_run: the mel inversion stores the magnitudes in the
loop kernel's own layout) unless --fused-mel 0 asks for gl_mel_to_mag + gl_run.

Workloads.  Every utterance is a function of its GLOBAL index only
(mel: numpy Generator seeded with (1234, id); start signal: elements
[id*L, (id+1)*L) of one device Philox stream), so any sharding sees the same
utterances:
  N = 1   BASELINE config 2: the batch of 64 (utterances 0..63) on one MI355X.
  N > 1   BASELINE config 3: ONE batch of 1024 utterances sharded per utterance,
          1024/N per rank, the same 1024 for every N ("scaling": "strong"); no
          collective on the data path (RCCL only for the barrier / max-over-ranks).

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel (the fused
Griffin-Lim iteration): algorithmic HBM bytes per launch / its average duration
measured with HIP events on the launch stream inside the timed region.
`cpu_baseline` times the float64 loop-form oracle (the reference's algorithm, dead
RMSE included) on one core ON THE GPU RUN'S OWN magnitudes and start signals
(utterances 0..3 of the batch), and `parity` compares the GPU waveforms of those
utterances with what it returned.  After the timed region (untimed for `value`) the same
run measures BASELINE's other configurations under `secondary`: config 1, config 5 through
the N1 entry point, config 4 (10 minutes, 100 iterations, held to the committed reference
windows) and a 128-utterance shard of config 3 -- ms per call, audio-s/s, the loop kernel's
time per iteration by HIP events, its algorithmic-bytes fraction, and parity against the oracle.
BENCH_FORCE_DIST=1 takes the torch.distributed / RCCL path with a world of one rank.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

SR = 22050
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec
VALU_FP32_PEAK_TF = 157.3 # MI355X_MICROARCH.md: packed fp32 vector peak (256 CUs x 256 flop/clk x 2.4 GHz)
CONFIG3_GLOBAL_BATCH = 1024
TOL = 1e-3                # north_star: max-abs error / peak and spectral-convergence delta


def synth_mels(B, T, seed):
    """exp(U), U ~ N(-2, 1.5^2) lightly smoothed over (time, band); one Generator for the batch
    (the tests' inputs)."""
    rng = np.random.default_rng(seed)
    u = rng.normal(-2.0, 1.5, size=(B, 80, T)).astype(np.float32)
    u = (u + np.roll(u, 1, axis=2) + np.roll(u, -1, axis=2)) / 3.0
    u = (u + np.roll(u, 1, axis=1) + np.roll(u, -1, axis=1)) / 3.0
    return np.exp(u).astype(np.float32)


def synth_mels_by_id(ids, T, seed=1234):
    """The same generator, one Generator PER UTTERANCE seeded with (seed, global id): utterance `id` is the same
    array whichever rank, shard or batch it is generated in."""
    out = np.empty((len(ids), 80, T), dtype=np.float32)
    for i, gid in enumerate(ids):
        out[i] = synth_mels(1, T, [seed, int(gid)])[0]
    return out


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def file_sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def cpu_baseline_and_parity(mags, x0s, gpu_x, N, H, iters, budget_s):
    """Reference algorithm (float64 numpy loop form incl. the dead RMSE, oracle/gl_oracle.py) on ONE core, on the
    GPU run's own magnitudes / start signals: utterance 0 always with all `iters` iterations, further utterances
    while the budget lasts.  Returns (cpu_baseline dict, parity dict)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gl_oracle as orc     # checker / baseline only -- never on the product path
    T = mags[0].shape[0]
    audio_s = (T * H + N) / SR
    orc.griffin_lim_loop(mags[0], N, H, 1, x0=x0s[0])          # page in numpy / pocketfft before the clock starts
    errs, scs, secs = [], [], []
    for u in range(len(mags)):
        if u > 0 and sum(secs) + secs[0] > budget_s:
            break
        t0 = time.perf_counter()
        ref = orc.griffin_lim_loop(mags[u], N, H, iters, x0=x0s[u])
        secs.append(time.perf_counter() - t0)
        got = gpu_x[u].astype(np.float64)
        errs.append(float(np.max(np.abs(got - ref)) / np.max(np.abs(ref))))
        m64 = mags[u].astype(np.float64)
        scs.append(abs(orc.spectral_convergence(got, m64, N, H) - orc.spectral_convergence(ref, m64, N, H)))
    n = len(secs)
    base = {"value": n * audio_s / sum(secs), "unit": "audio-s/s", "cores": 1, "kind": "port",
            "sample": f"{n} x 1 utterance of the GPU run's own batch (T={T}, n_fft {N}, hop {H}), all {iters} "
                      f"iterations each ({sum(secs):.1f} s timed); float64 numpy loop form of "
                      "audio_utilities.py:154-188 incl. the dead RMSE",
            "cpu_model": cpu_model(), "host_cpus": os.cpu_count()}
    parity = {"max_err_over_peak": max(errs), "sc_delta": max(scs), "utterances": list(range(n)),
              "batch": int(gpu_x.shape[0]) if hasattr(gpu_x, "shape") else len(gpu_x), "iters": iters,
              "tolerance": TOL, "ok": bool(max(errs) <= TOL and max(scs) <= TOL),
              "against": "float64 loop-form oracle on the same start signals and on the magnitudes the GPU's own mel "
                         "inversion produced (D2H after the timed region): the inversion's error is outside this figure; "
                         "it is held to 1e-6 of the peak against a float64 product in tests/test_gpu_parity.py"}
    return base, parity


def cpu_baseline_all_cores(N, H, T, iters, budget_s, per_utt_s):
    """the same loop form, one utterance per process, on every host core this job may use
    (16 per GPU on the pool's boxes; os.cpu_count() there reports the whole machine)."""
    cores = int(max(1, min(16, os.cpu_count() or 1)))
    reps = int(max(1, min(8, budget_s / max(per_utt_s, 1e-6))))
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
    worker = os.path.join(ROOT, "oracle", "cpu_worker.py")
    t0 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, worker, str(N), str(H), str(T), str(iters), str(reps), str(100 + i)],
                              stdout=subprocess.PIPE, env=env, text=True) for i in range(cores)]
    secs = []
    for pr in procs:
        out, _ = pr.communicate(timeout=600)
        if pr.returncode != 0:
            return {"error": f"worker exit {pr.returncode}"}
        secs.append(float(out.strip().splitlines()[-1]))
    wall = time.perf_counter() - t0
    audio_s = (T * H + N) / SR
    return {"value": cores * reps * audio_s / max(secs), "unit": "audio-s/s", "cores": cores, "kind": "port",
            "sample": f"{cores} processes x {reps} utterances x {iters} iterations, single-threaded each "
                      f"(slowest worker {max(secs):.1f} s, fastest {min(secs):.1f} s, {wall:.1f} s incl. start-up)"}


def traffic_for(lib_path, default_cfg, fused):
    """HBM bytes per launch of the dominant kernel from the PMC passes (FETCH_SIZE x2 gfx950 correction +
    WRITE_SIZE, separate rocprofv3 --pmc runs of this same command; tools_pmc_mem.sh), and the MFMA counters of the
    mel inversion on the bench path (tools_pmc_k1.sh).  Stored measurements: they are stamped only on the headline
    workload and only when they were taken with the very library this run loaded (sha256 recorded beside them);
    otherwise null with the reason.  Returns (traffic, note, k1 record or None)."""
    tpath = os.path.join(ROOT, "profiles", "r3_traffic.json")
    if not (default_cfg and fused):
        return None, "only measured for the headline workload on the fused path", None
    if not os.path.exists(tpath):
        return None, "profiles/r3_traffic.json absent", None
    with open(tpath) as fh:
        rec = json.load(fh)
    sha = file_sha256(lib_path)
    if rec.get("lib_sha256") != sha:
        return None, (f"profiles/r3_traffic.json was measured with libgl_hip.so sha256 {str(rec.get('lib_sha256'))[:12]}..., "
                      f"this run loaded {sha[:12]}...: not stamped"), None
    note = f"rocprofv3 PMC passes of this library (sha256 {sha[:12]}..., git {rec.get('git_head', '?')[:9]})"
    return rec.get("hbm_bytes_per_launch"), note, rec.get("k1")


def _events_ms(fn, reps):
    """Average GPU time of fn() over `reps` calls, HIP events on torch's current stream (the library's launch stream)."""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def _wall_ms(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def secondary_configs(dev, with_parity):
    """BASELINE's other configurations, measured after the headline's timed region.  Each entry:
    ms per call (wall, host launch included), whole-call audio-s/s, the Griffin-Lim loop's time per iteration by
    HIP events (`iter_us`), its algorithmic HBM bytes per iteration and `frac` of 8 TB/s, and -- with the CPU leg
    on -- parity of one utterance against the float64 oracle on the same magnitudes and start signal."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from tacotron2_amd import filters, griffin_lim
    from tacotron2_amd import audio_utilities as au
    from tacotron2_amd.plan import Plan
    out = {}

    def loop_stats(N, H, T, B, iter_ms):
        F = N // 2 + 1
        L = T * H + N
        algo = B * (T * F * 4 + 2 * L * 4)
        ach = algo / (iter_ms * 1e-3) / 1e9
        return {"iter_us": iter_ms * 1e3, "algorithmic_bytes_per_launch": algo, "achieved_GBps": ach,
                "frac": ach / HBM_PEAK_GBS}

    def oracle_parity(mag_u, x0_u, got_u, N, H, K):
        import gl_oracle as orc
        m64 = mag_u.astype(np.float64)
        ref = orc.griffin_lim_vec(m64, N, H, K, x0_u.astype(np.float64))
        got = got_u.astype(np.float64)
        return {"max_err_over_peak": float(np.max(np.abs(got - ref)) / np.max(np.abs(ref))),
                "sc_delta": float(abs(orc.spectral_convergence(got, m64, N, H) - orc.spectral_convergence(ref, m64, N, H))),
                "tolerance": TOL, "against": "float64 oracle (vectorised form) on the same magnitudes and start signal"}

    # ---- config 1: one ~8 s utterance, params.py defaults (n_fft 1024, hop 275), 60 iterations ------------------
    N, H, T, K, B = 1024, 275, 641, 60, 1
    L = T * H + N
    plan = Plan(N, H, T, B, dev)
    fb = torch.from_numpy(filters.mel(SR, N, 80, 125, 7600)).to(dev)
    mel = torch.from_numpy(synth_mels(B, T, 11)).to(dev)
    x0 = plan.randn((B, L), seed=7)
    mag = plan.mel_to_mag(fb, mel)
    x = torch.empty_like(x0)
    it_ms = _events_ms(lambda: plan.run(mag, x0, K, out=x), 20) / K
    call_ms = _wall_ms(lambda: (plan.mel_to_mag(fb, mel, out=mag), plan.run(mag, x0, K, out=x), plan.peak_normalise(x, mode=0)), 20)
    c1 = {"workload": "BASELINE config 1: one 8.0 s utterance, n_fft 1024 / hop 275 (params.py), mel->linear + "
                      "Griffin-Lim 60 iters + peak normalise", "batch": B, "frames": T, "iters": K,
          "ms_per_call": call_ms, "audio_s_per_s": B * L / SR / (call_ms * 1e-3), **loop_stats(N, H, T, B, it_ms)}
    if with_parity:
        plan.run(mag, x0, K, out=x)
        c1["parity"] = oracle_parity(mag[0].cpu().numpy(), x0[0].cpu().numpy(), x[0].cpu().numpy(), N, H, K)
    out["config1"] = c1
    plan.close()

    # ---- config 5: batch 32 of decoder outputs (T = 1001) through the N1 entry point ----------------------------
    T, B = 1001, 32
    L = T * H + N
    logmel = torch.from_numpy(np.ascontiguousarray(np.log(synth_mels(B, T, 55)).transpose(0, 2, 1))).to(dev)
    plan = Plan(N, H, T, B, dev)
    x0 = plan.randn((B, L), seed=8)

    def n1_call():
        return griffin_lim.logmel_to_audio(logmel, n_fft=N, hop=H, iterations=K, x_init=x0, return_mag=False)[0]

    call_ms = _wall_ms(n1_call, 8)
    mag = plan.logmel_to_mag(fb, logmel)
    x = torch.empty_like(x0)
    it_ms = _events_ms(lambda: plan.run(mag, x0, K, out=x), 8) / K
    c5 = {"workload": "BASELINE config 5: (32, 1001, 80) device-resident log-mel batch -> griffin_lim.logmel_to_audio("
                      "return_mag=False) (infer.py:24-29 glue: exp + transpose + mel->linear in one kernel, Griffin-Lim "
                      "60 iters, peak normalise), n_fft 1024 / hop 275",
          "batch": B, "frames": T, "iters": K, "ms_per_call": call_ms, "audio_s_per_s": B * L / SR / (call_ms * 1e-3),
          **loop_stats(N, H, T, B, it_ms)}
    if with_parity:
        got = griffin_lim.logmel_to_audio(logmel, n_fft=N, hop=H, iterations=K, x_init=x0, return_mag=False,
                                          normalize=None)[0]
        c5["parity"] = oracle_parity(mag[5].cpu().numpy(), x0[5].cpu().numpy(), got[5].cpu().numpy(), N, H, K)
    out["config5"] = c5
    plan.close()
    del logmel, mag, x, x0

    # ---- config 3's shard at 8 GPUs: 128 utterances per GPU, config-2 geometry ----------------------------------
    N, H, T, B = 2048, 256, 861, 128
    L = T * H + N
    plan = Plan(N, H, T, B, dev)
    fb = torch.from_numpy(filters.mel(SR, N, 80, 125, 7600)).to(dev)
    mel = torch.from_numpy(synth_mels_by_id(range(384, 512), T)).to(dev)      # rank 3's shard of the 1024
    x0 = plan.randn((B, L), seed=42, offset=384 * L)
    x = torch.empty_like(x0)
    call_ms = _events_ms(lambda: (plan.mel_run(fb, mel, x0, K, out=x), plan.peak_normalise(x, mode=0)), 4)
    out["shard128"] = {"workload": "config 3's per-GPU shard at 8 GPUs: 128 utterances (ids 384..511), config-2 geometry, "
                                   "mel->linear + Griffin-Lim 60 iters + peak normalise", "batch": B, "frames": T,
                       "iters": K, "ms_per_call": call_ms, "audio_s_per_s": B * L / SR / (call_ms * 1e-3),
                       **loop_stats(N, H, T, B, call_ms / K)}
    out["shard128"]["iter_us_note"] = "whole call / 60: includes 1/60 of the mel inversion and the normalisation"
    plan.close()
    del mel, x0, x

    # ---- config 4: one 10-minute utterance, 100 iterations, single piece ---------------------------------------
    import cfg4_inputs as c4
    T, K, B = c4.FRAMES, c4.ITERS, 1
    L = c4.signal_length()
    mag_h = c4.magnitudes(filters.mel(SR, N, 80, 125, 7600))
    mag = torch.from_numpy(mag_h).to(dev)[None]
    x0 = torch.from_numpy(c4.start_signal()).to(dev)[None]
    plan = Plan(N, H, T, B, dev)
    x = torch.empty_like(x0)
    call_ms = _events_ms(lambda: plan.run(mag, x0, K, out=x), 3)
    c4o = {"workload": "BASELINE config 4: one 600.1 s utterance (51 680 frames), Griffin-Lim 100 iters, single piece "
                       "on one GPU (input of oracle/cfg4_inputs.py; loop only, magnitudes resident)",
           "batch": B, "frames": T, "iters": K, "ms_per_call": call_ms, "audio_s_per_s": L / SR / (call_ms * 1e-3),
           **loop_stats(N, H, T, B, call_ms / K)}
    gpath = os.path.join(ROOT, "tests", "golden", "gl_cfg4_fullsize.npz")
    if with_parity and os.path.exists(gpath):
        d = np.load(gpath, allow_pickle=False)
        y = x[0].cpu().numpy()
        peak = float(d["peak100"])
        idx = d["starts"][:, None] + np.arange(c4.WIN)[None, :]
        werr = float(np.max(np.abs(y[idx].astype(np.float64) - d["win100"].astype(np.float64))) / peak)
        e = np.abs(c4.block_projections(y) - d["proj100"]) / np.sqrt(c4.PROJ_BLOCK) / peak
        e32 = np.abs(d["proj100_f32"] - d["proj100"]) / np.sqrt(c4.PROJ_BLOCK) / peak
        c4o["parity"] = {"windows_max_err_over_peak": werr, "windows": int(len(d["starts"])), "tolerance": TOL,
                         "blocks": int(len(e)), "over_tol_blocks": int((e > TOL).sum()), "worst_block": float(e.max()),
                         "median_block": float(np.median(e)),
                         "numpy_float32_over_tol_blocks": int((e32 > TOL).sum()), "numpy_float32_worst_block": float(e32.max()),
                         "against": "the reference's own run (tests/golden/gl_cfg4_fullsize.npz: 25 windows of 8 192 samples "
                                    "and one random-sign projection per 256-sample block of the whole signal); north_star's "
                                    "max-abs <= 1e-3 holds on the windows, NOT on every block of the 10 minutes: the blocks "
                                    "above it sit at the few spots where Griffin-Lim is locally unstable, where numpy's own "
                                    "float32 run of the reference algorithm misses it too (DESIGN section 2)"}
    out["config4"] = c4o
    plan.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--global-batch", type=int, default=0,
                    help="utterances in the whole job (default: 64 = config 2 on one GPU, 1024 = config 3 on several)")
    ap.add_argument("--batch", type=int, default=0, help="alias of --global-batch for single-GPU runs")
    ap.add_argument("--frames", type=int, default=861)
    ap.add_argument("--n-fft", type=int, default=2048)
    ap.add_argument("--hop", type=int, default=256)
    ap.add_argument("--iters", type=int, default=60)
    ap.add_argument("--path", type=int, default=0, help="0 fused (default), 1 generic kernels")
    ap.add_argument("--run-frames", type=int, default=0, help="pin the frames-per-wave-run of the fused kernels (0 = automatic)")
    ap.add_argument("--fused-mel", type=int, default=1,
                    help="1: mel inversion and loop as ONE call (gl_mel_run: packed magnitudes, no (B,T,F) array, no "
                         "repack pass) where the plan has it; 0: gl_mel_to_mag + gl_run")
    ap.add_argument("--graph", type=int, default=1, help="1: also time the step replayed from a captured hipGraph")
    ap.add_argument("--preroll-ms", type=float, default=40.0,
                    help="untimed steps before the W warmup steps until this much GPU time has passed: the card "
                         "idles through imports and set-up and its shader clock takes ~25 ms of work to ramp "
                         "(first iterations 165 us, steady state 135 us; abl/trace_var.sh)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline + parity leg; 0 disables")
    ap.add_argument("--secondary", type=int, default=1,
                    help="1: after the timed region also measure configs 1, 4, 5 and the 128-utterance shard (N = 1 only)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    all_cores = None
    force_dist = world == 1 and os.environ.get("BENCH_FORCE_DIST") == "1"
    if force_dist and args.cpu_seconds > 0:
        # child processes of the all-cores CPU leg start before this process touches the GPU
        all_cores = cpu_baseline_all_cores(args.n_fft, args.hop, args.frames, args.iters, args.cpu_seconds, 4.0)
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # BENCH_BACKEND=gloo rehearses the N>1 path on a box with fewer GPUs than ranks (ranks share cards)
        backend = os.environ.get("BENCH_BACKEND", "nccl")
        local_dev = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(local_dev)
        if force_dist:
            os.environ.setdefault("MASTER_PORT", "29581")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL prints a version banner on STDOUT when its communicator comes up; the contract is ONE JSON line there,
        # so stdout points at stderr until the first collective has run
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_dev))
            else:
                dist.init_process_group(backend)
            dist.barrier()
            if backend == "nccl":
                torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
    else:
        # the all-cores CPU leg starts child processes: run it before this process touches the GPU
        if args.cpu_seconds > 0:
            all_cores = cpu_baseline_all_cores(args.n_fft, args.hop, args.frames, args.iters, args.cpu_seconds, 4.0)
        torch.cuda.set_device(0)
    n_gpus = world if world > 1 else 1
    if args.gpus != n_gpus and rank == 0:
        print(f"note: --gpus {args.gpus} but WORLD_SIZE={world}; using {n_gpus}", file=sys.stderr)
    dev = torch.device("cuda", torch.cuda.current_device())

    from tacotron2_amd import _lib, filters
    from tacotron2_amd.plan import Plan
    from tacotron2_amd.sharding import shard_bounds

    N, H, T, K = args.n_fft, args.hop, args.frames, args.iters
    G = args.global_batch or args.batch or (CONFIG3_GLOBAL_BATCH if n_gpus > 1 else 64)
    lo, hi = shard_bounds(G, n_gpus, rank)               # this rank's utterances of the ONE global batch
    B = hi - lo
    F = N // 2 + 1
    L = T * H + N
    plan = Plan(N, H, T, B, dev)
    plan.set_path(args.path)
    plan.set_run_frames(args.run_frames)
    fb = torch.from_numpy(filters.mel(SR, N, 80, 125, 7600)).to(dev)
    mel = torch.from_numpy(synth_mels_by_id(range(lo, hi), T)).to(dev)
    x0 = plan.randn((B, L), seed=42, offset=lo * L)      # resident start signal, global-index consistent
    mag = torch.empty((B, T, F), dtype=torch.float32, device=dev)
    x = torch.empty((B, L), dtype=torch.float32, device=dev)

    gl_ms = []
    one_call = bool(args.fused_mel) and plan.mel_run_supported()

    def mel_and_loop(timed=False):
        """griffin_lim.py:18-24 for the batch.  The events bracket the K loop launches (torch's current stream is the
        stream the library launches on); with the one-call form they also take in the mel inversion, without it
        the repack pass -- either way 1/K of a ~0.1 ms kernel on top of each iteration."""
        if not one_call:
            plan.mel_to_mag(fb, mel, out=mag)
        if timed:
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
        if one_call:
            plan.mel_run(fb, mel, x0, K, out=x)
        else:
            plan.run(mag, x0, K, out=x)
        if timed:
            e1.record()
            gl_ms.append((e0, e1))

    def step(timed):
        mel_and_loop(timed)
        plan.peak_normalise(x, mode=0)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(v):
        if dist is None:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    preroll_steps = 0
    tp = time.perf_counter()
    while (time.perf_counter() - tp) * 1e3 < args.preroll_ms:
        step(False)
        torch.cuda.synchronize()
        preroll_steps += 1
    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)

    # same step replayed from one captured hipGraph (64 kernels, one submission); reported beside
    # the stream-launched number, which stays `value` (its kernels are the ones timed for `roofline`)
    graph_ms = None
    if args.graph:
        peaks = torch.empty((B,), dtype=torch.float32, device=dev)
        with plan.capture() as g:
            mel_and_loop()
            plan.peak_normalise(x, mode=0, peaks=peaks)
        x_stream = x.clone()
        for _ in range(max(1, args.warmup)):
            g.launch()
        barrier()
        tg = time.perf_counter()
        for _ in range(args.steps):
            g.launch()
        barrier()
        graph_ms = max_over_ranks((time.perf_counter() - tg) / args.steps * 1e3)
        assert torch.equal(x, x_stream), "graph replay differs from the stream-launched step"

    finite = bool(torch.isfinite(x).all().item())
    rccl = None
    if force_dist:
        # the optional final gather (north_star: "no RCCL collective except a final gather") on the real stack with the
        # one rank a one-GPU box has: the padded all_gather of sharding.gather_waveforms right behind the library's
        # kernels, no synchronisation in between
        from tacotron2_amd.sharding import gather_waveforms
        step(False)
        ref_x = x.clone()
        tg0 = time.perf_counter()
        full = gather_waveforms(x, B, always_collective=True)
        torch.cuda.synchronize()
        rccl = {"backend": dist.get_backend(), "world": dist.get_world_size(), "gather_equals_local": bool(torch.equal(full, ref_x)),
                "gather_ms_first_call": (time.perf_counter() - tg0) * 1e3,
                "note": "world of ONE rank: proves init, barrier / all_reduce and the gather on RCCL; N > 1 is unmeasured"}
        del full, ref_x
    if rank == 0:
        audio_s_total = G * L / SR * args.steps
        value = audio_s_total / dt
        # dominant kernel: one fused GL iteration over the rank's B utterances
        iter_ms = sum(a.elapsed_time(b) for a, b in gl_ms) / (len(gl_ms) * K)
        algo_bytes = B * (T * F * 4 + 2 * L * 4)          # read mag + read x + write x
        achieved = algo_bytes / (iter_ms * 1e-3) / 1e9
        # SURVEY 8d flop count of one iteration (radix-2 convention 2.5 N log2 N per real FFT, windows,
        # projection at 8 flop/bin, overlap-add): the fused loop sits at the HBM/VALU ridge, so both are reported
        algo_flops = B * (2 * T * 2.5 * N * np.log2(N) + 2 * T * N + 8 * T * F + T * N)
        valu_tf = algo_flops / (iter_ms * 1e-3) / 1e12
        geom = (N, H, T, K) == (2048, 256, 861, 60)
        default_cfg = geom and n_gpus == 1 and G == 64
        if default_cfg:
            workload = ("BASELINE config 2: batch of 64 synthetic 10 s 80-band mels on one MI355X, "
                        "mel->linear + Griffin-Lim 60 iters + peak normalise")
        elif geom and n_gpus > 1 and G == CONFIG3_GLOBAL_BATCH:
            workload = (f"BASELINE config 3: ONE batch of 1024 synthetic 10 s 80-band mels sharded per utterance across "
                        f"{n_gpus} MI355X ({B} on rank 0), mel->linear + Griffin-Lim 60 iters + peak normalise")
        else:
            workload = (f"non-headline geometry: batch of {G} synthetic 80-band mels over {n_gpus} GPU(s), n_fft {N}, "
                        f"hop {H}, {T} frames, mel->linear + Griffin-Lim {K} iters + peak normalise")
        traffic, traffic_note, k1 = traffic_for(_lib.LIB_PATH, default_cfg, args.path == 0)
        out = {
            "metric": f"audio-seconds reconstructed/sec (GL {K} iters, n_fft {N})",
            "value": value, "unit": "audio-s/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "preroll_steps": preroll_steps, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if n_gpus > 1 else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload,
                       "n_fft": N, "hop": H, "frames": T, "iters": K, "batch_per_gpu": B,
                       "global_batch": G, "audio_s_per_utt": L / SR,
                       "parallelism": f"utterance-sharded x{n_gpus}, no data-path collective",
                       "kernel_path": "fused" if args.path == 0 else "generic",
                       "mel_inversion": "gl_mel_run (packed magnitudes, one call)" if one_call else "gl_mel_to_mag + gl_run"},
            "x_realtime_per_gpu": value / n_gpus,
            "roofline": {"bound": "hbm", "kernel": "fused Griffin-Lim iteration (one launch per iteration)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note,
                         "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": iter_ms,
                         "valu_fp32": {"achieved": valu_tf, "peak": VALU_FP32_PEAK_TF, "unit": "TFLOP/s",
                                       "frac": valu_tf / VALU_FP32_PEAK_TF,
                                       "algorithmic_flops_per_launch": float(algo_flops)}},
            "outputs_finite": finite,
            "hipgraph_ms_per_step": graph_ms,
        }
        if k1 is not None and one_call and "avg_us" in k1:
            c = k1.get("counters_per_launch", {})
            out["roofline_k1"] = {
                "bound": "mfma", "kernel": "k_mel_to_mag<false, true, true> (mel -> linear magnitudes in the loop's packed "
                                           "layout, inside gl_mel_run; 1 of the step's 63 launches)",
                "achieved": k1["achieved_TFLOPs"], "peak": VALU_FP32_PEAK_TF, "unit": "TFLOP/s", "frac": k1["frac"],
                "avg_launch_us": k1["avg_us"], "useful_flops_per_launch": k1["useful_flops_per_launch"],
                "mfma_flops_issued_per_launch": k1.get("mfma_flops_issued_per_launch"),
                "mfma_busy_share_of_simd_time": (c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * c["SQ_BUSY_CU_CYCLES"])
                                               if c.get("SQ_BUSY_CU_CYCLES") else None),
                "source": "stored rocprofv3 record of this library (kernel trace + --pmc pass, profiles/r3_traffic.json: "
                          "k1); fp32-input MFMA peak = the fp32 vector peak on gfx950"}
        if rccl is not None:
            out["rccl"] = rccl
        if args.cpu_seconds > 0:
            # un-normalised Griffin-Lim output of the batch once more (the timed step normalises in place), then the
            # first utterances back to the host for the CPU leg: 4 at N = 1 (they are also the CPU baseline), the
            # first 2 of rank 0's shard at N > 1 (parity only: the baseline is an N = 1 figure)
            nu = min(4 if n_gpus == 1 else 2, B)
            mel_and_loop()                                 # the measured path's samples ...
            plan.mel_to_mag(fb, mel, out=mag)              # ... and the magnitudes the oracle starts from
            torch.cuda.synchronize()
            mags = [mag[u].cpu().numpy() for u in range(nu)]
            x0s = [x0[u].cpu().numpy().astype(np.float64) for u in range(nu)]
            gx = x[:nu].cpu().numpy()
            base, parity = cpu_baseline_and_parity(mags, x0s, gx, N, H, K, args.cpu_seconds)
            parity["batch"] = B
            parity["utterances"] = [lo + u for u in parity["utterances"]]
            out["parity"] = parity
            if n_gpus == 1:
                out["cpu_baseline"] = base
                if all_cores is not None:
                    out["cpu_baseline"]["all_cores"] = all_cores
        if args.secondary and n_gpus == 1:
            del mag, mel, x0, x
            plan.close()
            t_sec = time.perf_counter()
            out["secondary"] = secondary_configs(dev, args.cpu_seconds > 0)
            out["secondary"]["seconds"] = time.perf_counter() - t_sec
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
