#!/bin/bash
# kernel-trace stats of the config 5 / config 1 timing script -> gpurun_out/prof_cfg5_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc5
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc5 -- python3 /root/repo/abl/cfg5.py > /tmp/pc5.log 2>&1 || { tail -5 /tmp/pc5.log; exit 1; }
grep config /tmp/pc5.log > /root/repo/gpurun_out/prof_cfg5.txt
cp $(ls /tmp/pc5/*/*kernel_stats.csv | head -1) /root/repo/gpurun_out/prof_cfg5_kernel_stats.csv
cat /root/repo/gpurun_out/prof_cfg5.txt; cut -c1-150 /root/repo/gpurun_out/prof_cfg5_kernel_stats.csv | head -8
