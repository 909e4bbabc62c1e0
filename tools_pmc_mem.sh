#!/bin/bash
# usage: tools_pmc_mem.sh <tag> ; memory-side PMC passes of one bench step -> gpurun_out/pmcmem_<tag>_<n>.txt
tag=$1
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "FETCH_SIZE TCC_REQ_sum" \
            "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
            "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum" \
            "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/pmcm_$i
  timeout -s KILL 90 rocprofv3 --pmc $ctrs --output-format csv -d /tmp/pmcm_$i -- python3 /root/repo/bench.py --steps 1 --warmup 0 --preroll-ms 0 --cpu-seconds 0 --graph 0 --secondary 0 > /tmp/pmcm_$i.log 2>&1 || { echo "pass $i failed"; tail -5 /tmp/pmcm_$i.log; exit 1; }
  f=$(ls /tmp/pmcm_$i/*/*counter_collection.csv | head -1)
  python3 - "$f" <<'PY' > /root/repo/gpurun_out/pmcmem_${tag}_$i.txt
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if "regring" in k or "fast2048" in k or "generic_gl" in k or "pair" in k:
        for c, v in d.items():
            print(k, c, "per-dispatch", v / cnt[(k, c)], "dispatches", cnt[(k, c)])
PY
  cat /root/repo/gpurun_out/pmcmem_${tag}_$i.txt
done
