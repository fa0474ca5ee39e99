#!/bin/bash
# tools/build_variant.sh <name> [-DFLAG ...] — build libchipvideo with extra defines into variants/<name>.so
# (A/B experiments on the GPU box: cp variants/<name>.so swiftvideo_amd/libchipvideo.so; variants/ is git-ignored)
set -e
NAME=$1; shift
cd "$(dirname "$0")/../swiftvideo_amd/csrc"
mkdir -p ../../variants/obj_$NAME
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -w"
for f in chipvideo.cpp kernels_general.hip.cpp kernels_fast.hip.cpp kernels_fast_rgb.hip.cpp kernels_lanczos.hip.cpp; do
  /opt/rocm/bin/hipcc $FLAGS "$@" -x hip -c $f -o ../../variants/obj_$NAME/${f%.cpp}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/$NAME.so ../../variants/obj_$NAME/*.o -lhiprtc
rm -rf ../../variants/obj_$NAME
