#!/bin/bash
for g in "$@"; do
  timeout -k 10 120 python bench.py --run-frames $g --cpu-seconds 0 --steps 5 --warmup 2 --graph 0 > /tmp/g_$g.json || exit 1
  python - <<PY
import json; d=json.load(open("/tmp/g_$g.json")); print("G=$g", d["roofline"]["avg_launch_ms"], d["value"])
PY
done
