#!/bin/bash
# duration of the dominant kernel by its position inside a step (iteration 0..59), from a kernel trace
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/itp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/itp -- python3 /root/repo/bench.py --cpu-seconds 0 --graph 0 --steps 10 --warmup 2 > /tmp/itp.log 2>&1 || { tail -5 /tmp/itp.log; exit 1; }
f=$(ls /tmp/itp/*/*kernel_trace.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
steps = []; cur = []
for r in rows:
    if "regring" in r["Kernel_Name"]:
        cur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    elif cur:
        steps.append(cur); cur = []
if cur: steps.append(cur)
steps = [s for s in steps if len(s) == 60][-10:]
import statistics
for i in range(60):
    v = [s[i] for s in steps]
    print(i, round(statistics.mean(v) / 1e3, 1), round(min(v) / 1e3, 1), round(max(v) / 1e3, 1))
PY
