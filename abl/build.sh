#!/bin/bash
# abl/build.sh <name> [extra hipcc flags...] -> abl/libgl_abl<name>.so (experiment library, selected with GL_HIP_LIB)
name=$1; shift
cd /root/repo/tacotron2_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-unused-function -ffp-contract=fast "$@" \
  -Xclang -target-feature -Xclang -load-store-opt \
  -shared -o /root/repo/abl/libgl_abl$name.so gl_api.hip gl_generic.hip gl_fast.hip gl_fast2048.hip gl_fast1024.hip gl_mel.hip gl_misc.hip 2>&1 | grep -v "not a recognized feature"
