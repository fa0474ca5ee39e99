#!/bin/bash
# tools/kstats.sh <stem> — VGPR / spill / LDS / scratch numbers of every kernel in swiftvideo_amd/csrc/<stem>.hip.o
set -e
LLVM=/opt/rocm/lib/llvm/bin
obj=swiftvideo_amd/csrc/$1.hip.o
[ -f "$2" ] && obj=$2
tmp=$(mktemp -d)
$LLVM/llvm-objcopy --dump-section=.hip_fatbin=$tmp/f.fatbin $obj
$LLVM/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$tmp/f.fatbin --output=$tmp/k.co
$LLVM/llvm-readelf --notes $tmp/k.co | grep -E "\.name:|vgpr_count|vgpr_spill|sgpr_count|sgpr_spill|private_segment_fixed|group_segment_fixed" | sed 's/^ *//' | paste - - - - - - - | sed 's/\t/ /g'
[ -n "$KEEP_CO" ] && cp $tmp/k.co $KEEP_CO
rm -rf $tmp
