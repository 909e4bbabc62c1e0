#!/bin/bash
# usage: tools_pmc.sh <tag> ; runs 3 PMC passes of one bench step, writes gpurun_out/pmc_<tag>_<n>.csv
tag=$1
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
            "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_IFETCH SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout -s KILL 90 rocprofv3 --pmc $ctrs --output-format csv -d /tmp/pmc_$i -- python3 /root/repo/bench.py --steps 1 --warmup 0 --preroll-ms 0 --cpu-seconds 0 --graph 0 --secondary 0 > /tmp/pmc_$i.log 2>&1 || { echo "pass $i failed"; tail -5 /tmp/pmc_$i.log; exit 1; }
  f=$(ls /tmp/pmc_$i/*/*counter_collection.csv | head -1)
  python3 - "$f" <<'PY' > /root/repo/gpurun_out/pmc_${tag}_$i.txt
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if "regring" in k or "fast2048" in k or "generic_gl" in k:
        for c, v in d.items():
            print(k, c, "per-dispatch", v / cnt[(k, c)], "dispatches", cnt[(k, c)])
PY
  cat /root/repo/gpurun_out/pmc_${tag}_$i.txt
done
