#!/bin/bash
# tools/build_variant.sh <name> <file.hip.cpp|all> [-DFLAG ...] — build a libchipvideo variant into variants/<name>.so:
# <file> (or every source with `all`) is recompiled with the extra defines, the rest is taken from the in-tree objects
# (make -C swiftvideo_amd/csrc first).  A/B on the GPU box: CHV_LIB=variants/<name>.so python bench.py ...  (variants/ is git-ignored)
set -e
NAME=$1; FILE=$2; shift 2
cd "$(dirname "$0")/../swiftvideo_amd/csrc"
OBJ=../../variants/obj_$NAME
mkdir -p $OBJ
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -w -DCHV_ARCH=\"gfx950\" -DCHV_HIPCC_VERSION=\"variant\""
for f in chipvideo.cpp kernels_*.hip.cpp; do
  if [ "$FILE" = all ] || [ "$FILE" = "$f" ]; then /opt/rocm/bin/hipcc $FLAGS "$@" -x hip -c $f -o $OBJ/${f%.cpp}.o &
  else cp ${f%.cpp}.o $OBJ/; fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/$NAME.so $OBJ/*.o -lhiprtc
rm -rf $OBJ
