#!/bin/bash
# usage: tools_pmc_k1.sh <tag> ; MFMA counters of the mel inversion ON THE BENCH PATH (k_mel_to_mag<false, true, true>,
# the PACKED instance inside gl_mel_run) from one bench step -> gpurun_out/pmck1_<tag>.txt
tag=$1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmck1
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmck1 -- python3 /root/repo/bench.py --steps 1 --warmup 0 --preroll-ms 0 --cpu-seconds 0 --graph 0 --secondary 0 > /tmp/pmck1.log 2>&1 || { echo "K1 pass failed"; tail -5 /tmp/pmck1.log; exit 1; }
f=$(ls /tmp/pmck1/*/*counter_collection.csv | head -1)
python3 - "$f" <<'PY' > /root/repo/gpurun_out/pmck1_${tag}.txt
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:90]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if "mel_to_mag" in k:
        for c, v in d.items():
            print(k, c, "per-dispatch", v / cnt[(k, c)], "dispatches", cnt[(k, c)])
PY
cat /root/repo/gpurun_out/pmck1_${tag}.txt
