#!/bin/bash
# usage: tools_profile.sh <tag>: kernel-trace stats of the default bench run -> gpurun_out/prof_<tag>/
tag=$1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python3 /root/repo/bench.py --cpu-seconds 0 --graph 0 --secondary 0 > /tmp/prof_$tag.json 2> /tmp/prof_$tag.err || { tail -5 /tmp/prof_$tag.err; exit 1; }
mkdir -p /root/repo/gpurun_out/prof_$tag
cp /tmp/prof_$tag.json /root/repo/gpurun_out/prof_$tag/bench.json
cp $(ls /tmp/prof_$tag/*/*kernel_stats.csv | head -1) /root/repo/gpurun_out/prof_$tag/kernel_stats.csv
cat /root/repo/gpurun_out/prof_$tag/kernel_stats.csv | cut -c1-200
