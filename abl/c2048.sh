#!/bin/bash
# small problems at n_fft 2048 / hop 256: per-iteration time for a few (batch, frames) sizes
for bt in "$@"; do
  b=${bt%x*}; t=${bt#*x}
  timeout -k 10 200 python bench.py --batch $b --frames $t --cpu-seconds 0 --graph 0 > /tmp/c2_$bt.json 2>/dev/null || exit 1
  python - <<PY
import json
d=json.load(open("/tmp/c2_$bt.json")); r=d["roofline"]; print("B x T = $bt", "audio-s/s", round(d["value"]), "ms/step", round(d["ms_per_step"],3), "us/iter", round(r["avg_launch_ms"]*1e3,1))
PY
done
