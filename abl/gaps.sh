#!/bin/bash
# launch-to-launch gaps of the dominant kernel from a kernel trace of one short bench run
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gaps
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/gaps -- python3 /root/repo/bench.py --cpu-seconds 0 --graph 0 --steps 3 --warmup 1 > /tmp/gaps.log 2>&1 || { tail -5 /tmp/gaps.log; exit 1; }
f=$(ls /tmp/gaps/*/*kernel_trace.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
gaps = []; durs = []
for a, b in zip(rows, rows[1:]):
    if "regring" in a["Kernel_Name"] and "regring" in b["Kernel_Name"]:
        gaps.append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
        durs.append(int(b["End_Timestamp"]) - int(b["Start_Timestamp"]))
gaps.sort(); durs.sort()
n = len(gaps)
print("pairs", n, "gap ns: min", gaps[0], "median", gaps[n // 2], "p90", gaps[int(n * 0.9)], "max", gaps[-1])
print("duration ns: min", durs[0], "median", durs[n // 2], "p90", durs[int(n * 0.9)])
PY
