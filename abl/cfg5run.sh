#!/bin/bash
for a in "$@"; do
  lib=/root/repo/abl/libgl_abl$a.so
  [ "$a" = "base" ] && lib=/root/repo/tacotron2_amd/libgl_hip.so
  echo "variant $a"; GL_HIP_LIB=$lib timeout -k 10 120 python abl/cfg5.py 2>&1 | grep "config 5" || exit 1
done
