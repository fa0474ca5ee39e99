#!/bin/bash
# tools/build_head_variant.sh <name> <file.hip.cpp> [more files] — variants/<name>.so = the in-tree objects with the named sources compiled from their COMMITTED (HEAD) text:
# the "before" side of an A/B on an uncommitted change to one kernel source (CHV_LIB=variants/<name>.so).  Headers are the working tree's.
set -e
NAME=$1; shift
cd "$(dirname "$0")/../swiftvideo_amd/csrc"
OBJ=../../variants/obj_$NAME
mkdir -p $OBJ
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -w -DCHV_ARCH=\"gfx950\" -DCHV_HIPCC_VERSION=\"variant\""
for f in chipvideo.cpp kernels_*.hip.cpp; do cp ${f%.cpp}.o $OBJ/; done
for FILE in "$@"; do
  git show HEAD:swiftvideo_amd/csrc/$FILE > head_tmp_$FILE
  /opt/rocm/bin/hipcc $FLAGS -x hip -c head_tmp_$FILE -o $OBJ/${FILE%.cpp}.o || { rm -f head_tmp_$FILE; exit 1; }
  rm -f head_tmp_$FILE
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/$NAME.so $OBJ/*.o -lhiprtc
rm -rf $OBJ
