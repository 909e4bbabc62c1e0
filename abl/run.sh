#!/bin/bash
# times bench steps per experiment library (some variants compute wrong numbers by construction; timing only)
for a in "$@"; do
  lib=/root/repo/abl/libgl_abl$a.so
  [ "$a" = "base" ] && lib=/root/repo/tacotron2_amd/libgl_hip.so
  GL_HIP_LIB=$lib timeout -k 10 120 python bench.py --cpu-seconds 0 --steps 10 --warmup 2 --graph 0 --secondary 0 > /tmp/abl_$a.json || exit 1
  python - <<PY
import json; d=json.load(open("/tmp/abl_$a.json")); print("variant $a", d["roofline"]["avg_launch_ms"], "finite", d["outputs_finite"])
PY
done
