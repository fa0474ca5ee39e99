#!/bin/bash
# tools/gpu_iter.sh <tag> <variants...> — one GPU-box iteration: the -m gpu suite, then the headline workload on the in-tree
# library and on each variants/<v>.so (A/B), everything summarised into gpurun_out/iter_<tag>.txt
TAG=$1; shift
OUT=gpurun_out/iter_$TAG.txt
mkdir -p gpurun_out
{
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -12
for v in default "$@"; do
  [ "$v" = default ] && unset CHV_LIB || export CHV_LIB=variants/$v.so
  for wl in ${WLS:-pipeline}; do
  timeout 600 python bench.py --workload $wl --also none --no-cpu-baseline --min-seconds 0.6 --steps 10 --warmup 3 2>&1 | tail -1 | \
    python -c "import sys,json
try:
    d=json.loads(sys.stdin.read()); print('$v $wl', d['config']['kernel'], 'launch_ms', round(d['roofline']['launch_ms'],4), 'frac', round(d['roofline']['frac'],4), 'Gpix/s', round(d['value'],1), 'verified', d['config']['verified_vs_oracle'])
except Exception as e: print('$v $wl FAILED', e)"
  done
done
unset CHV_LIB
} > $OUT 2>&1
cat $OUT
