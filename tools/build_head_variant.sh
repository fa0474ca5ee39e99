#!/bin/bash
# tools/build_head_variant.sh <name> <file.hip.cpp> — variants/<name>.so = the in-tree objects with <file> compiled from its COMMITTED (HEAD) text:
# the "before" side of an A/B on an uncommitted change to one kernel source (CHV_LIB=variants/<name>.so).  Headers are the working tree's.
set -e
NAME=$1; FILE=$2
cd "$(dirname "$0")/../swiftvideo_amd/csrc"
OBJ=../../variants/obj_$NAME
mkdir -p $OBJ
git show HEAD:swiftvideo_amd/csrc/$FILE > head_tmp_$FILE
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -w -DCHV_ARCH=\"gfx950\" -DCHV_HIPCC_VERSION=\"variant\""
/opt/rocm/bin/hipcc $FLAGS -x hip -c head_tmp_$FILE -o $OBJ/${FILE%.cpp}.o || { rm -f head_tmp_$FILE; exit 1; }
rm -f head_tmp_$FILE
for f in chipvideo.cpp kernels_*.hip.cpp; do [ "$f" = "$FILE" ] || cp ${f%.cpp}.o $OBJ/; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/$NAME.so $OBJ/*.o -lhiprtc
rm -rf $OBJ
