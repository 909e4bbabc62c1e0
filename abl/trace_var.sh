#!/bin/bash
# per-dispatch durations of the iteration kernel across one bench run (where does the spread come from?)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tv
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tv -- python3 /root/repo/bench.py --cpu-seconds 0 --graph 0 > /tmp/tv.json 2> /tmp/tv.err || { tail -5 /tmp/tv.err; exit 1; }
f=$(ls /tmp/tv/*/*kernel_trace.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'regring' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
gaps=[(int(rows[i+1]['Start_Timestamp'])-int(rows[i]['End_Timestamp']))/1e3 for i in range(len(rows)-1)]
print("n", len(d))
for s in range(0, len(d), 60):
    blk=d[s:s+60]
    print("step", s//60, "first5", [round(x,1) for x in blk[:5]], "mean", round(sum(blk)/len(blk),1), "min", round(min(blk),1), "max", round(max(blk),1), "last3", [round(x,1) for x in blk[-3:]])
g=[x for x in gaps if x < 50]
print("gap between consecutive iteration kernels: mean us", sum(g)/len(g), "max", max(g))
PY
