#!/bin/bash
# times one bench step set per ablation library (numbers are wrong by construction; timing only)
for a in "$@"; do
  GL_HIP_LIB=/root/repo/abl/libgl_abl$a.so timeout -k 10 120 python bench.py --cpu-seconds 0 --steps 3 --warmup 1 > /tmp/abl_$a.json || exit 1
  python - <<PY
import json; d=json.load(open("/tmp/abl_$a.json")); print("variant $a", d["roofline"]["avg_launch_ms"])
PY
done
