#!/bin/bash
for v in "$@"; do
echo "== $v"
GL_HIP_LIB=/root/repo/abl/libgl_abl$v.so timeout -k 10 120 python bench.py --cpu-seconds 0 --steps 1 --warmup 0 --graph 0 2>&1 | grep -v amdgpu.ids | cut -c1-100 | grep -v metric
done
