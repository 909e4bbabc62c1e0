#!/bin/bash
# single-utterance latency (config 1 geometry and a few batch sizes), stream-launched
for b in "$@"; do
  timeout -k 10 200 python bench.py --batch $b --frames 641 --n-fft 1024 --hop 275 --cpu-seconds 0 --graph 0 > /tmp/c1_$b.json 2>/dev/null || exit 1
  python - <<PY
import json
d=json.load(open("/tmp/c1_$b.json")); r=d["roofline"]; print("B=$b", "audio-s/s", round(d["value"]), "ms/step", round(d["ms_per_step"],3), "us/iter", round(r["avg_launch_ms"]*1e3,1))
PY
done
