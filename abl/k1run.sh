#!/bin/bash
# times the mel inversion kernel per experiment library
for a in "$@"; do
  lib=/root/repo/abl/libgl_abl$a.so
  [ "$a" = "base" ] && lib=/root/repo/tacotron2_amd/libgl_hip.so
  echo "variant $a"
  GL_HIP_LIB=$lib timeout -k 10 120 python abl/k1bench.py 2>&1 | grep mel_to_mag || exit 1
done
