#!/bin/bash
# tools/gpu_ab.sh — timing-only A/B lines: "<label> <env assignments> -- <workloads>" per line of $1 (default tools/ab.list)
mkdir -p gpurun_out
LIST=${1:-tools/ab.list}
{
while IFS= read -r line; do
  [ -z "$line" ] && continue
  label=${line%% *}; rest=${line#* }; envs=${rest%%--*}; wls=${rest#*--}
  for wl in $wls; do
    env $envs timeout 600 python bench.py --workload $wl --also none --no-cpu-baseline --min-seconds ${AB_SECONDS:-0.5} --steps 10 --warmup 3 $AB_ARGS 2>&1 | tail -1 | \
    python -c "import sys,json
try:
    d=json.loads(sys.stdin.read()); print('$label $wl', d['config']['kernel'], 'launch_ms', round(d['roofline']['launch_ms'],4), 'frac', round(d['roofline']['frac'],4), 'verified', d['config']['verified_vs_oracle'])
except Exception as e: print('$label $wl FAILED', e)"
  done
done < $LIST
} > gpurun_out/ab.txt 2>&1
cat gpurun_out/ab.txt
