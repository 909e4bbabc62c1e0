#!/bin/bash
# config 1 (one utterance, n_fft 1024 / hop 275): kernel durations and gaps between consecutive iteration kernels
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tc1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tc1 -- python3 /root/repo/bench.py --batch 1 --frames 641 --n-fft 1024 --hop 275 --cpu-seconds 0 --graph 0 --steps 3 > /tmp/tc1.json 2> /tmp/tc1.err || { tail -5 /tmp/tc1.err; exit 1; }
f=$(ls /tmp/tc1/*/*kernel_trace.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
it=[i for i,r in enumerate(rows) if 'gl1024' in r['Kernel_Name']]
d=[(int(rows[i]['End_Timestamp'])-int(rows[i]['Start_Timestamp']))/1e3 for i in it]
g=[(int(rows[i+1]['Start_Timestamp'])-int(rows[i]['End_Timestamp']))/1e3 for i in it[:-1] if 'gl1024' in rows[i+1]['Kernel_Name']]
print(rows[it[-1]]['Kernel_Name'][:60], "n", len(d))
d2=d[len(d)//2:]; g2=g[len(g)//2:]
print("kernel us: mean", sum(d2)/len(d2), "min", min(d2), "max", max(d2))
print("gap us: mean", sum(g2)/len(g2), "min", min(g2), "max", max(g2))
PY
