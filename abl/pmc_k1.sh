#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmck1
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d /tmp/pmck1 -- python3 /root/repo/abl/k1bench.py > /tmp/pmck1.log 2>&1 || { tail -5 /tmp/pmck1.log; exit 1; }
f=$(ls /tmp/pmck1/*/*counter_collection.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:40]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if "mel" in k:
        for c, v in d.items(): print(k, c, v / cnt[(k, c)])
PY
