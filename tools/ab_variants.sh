#!/bin/bash
# tools/ab_variants.sh <bench args> -- v1 v2 ... : bench each variants/<v>.so (run on the GPU box)
ARGS=()
while [ "$1" != "--" ]; do ARGS+=("$1"); shift; done; shift
cp swiftvideo_amd/libchipvideo.so /tmp/lib_orig.so
for round in 1 2; do
for v in "$@"; do
  cp variants/$v.so swiftvideo_amd/libchipvideo.so
  python bench.py "${ARGS[@]}" --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), 'ms', round(d['roofline']['frac'],4))"
done; done
cp /tmp/lib_orig.so swiftvideo_amd/libchipvideo.so
