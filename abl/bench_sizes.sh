#!/bin/bash
# headline bench plus the per-GPU shards of config 3 (B=128) and a larger batch
timeout -k 10 300 python bench.py > gpurun_out/bench_r1e.json 2> gpurun_out/bench_r1e.err || { tail -5 gpurun_out/bench_r1e.err; exit 1; }
timeout -k 10 200 python bench.py --batch 128 --cpu-seconds 0 --graph 0 > gpurun_out/bench_b128.json 2>/dev/null || exit 1
timeout -k 10 200 python bench.py --batch 256 --cpu-seconds 0 --graph 0 > gpurun_out/bench_b256.json 2>/dev/null || exit 1
python - <<PY
import json
for f in ["gpurun_out/bench_r1e.json","gpurun_out/bench_b128.json","gpurun_out/bench_b256.json"]:
    d=json.load(open(f)); r=d["roofline"]; print(f, round(d["value"]), round(d["ms_per_step"],3), round(r["avg_launch_ms"]*1e3,1), round(r["frac"],4), round(r["valu_fp32"]["frac"],4), d.get("cpu_baseline"))
PY
