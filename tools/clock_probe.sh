#!/bin/bash
# tools/clock_probe.sh <label> <env assignments...> -- <workload>: run one workload for ~6 s and sample the GPU's clocks and power twice a second
# (is a kernel that is neither at its instruction floor nor at its memory floor held back by the power limit?)
label=$1; shift
envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
wl=$1
mkdir -p gpurun_out
env "${envs[@]}" python bench.py --workload $wl --also none --no-cpu-baseline --no-route-regret --min-seconds 6 --steps 10 --warmup 3 > gpurun_out/clk_$label.json 2>/dev/null &
pid=$!
sleep 2.5
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr -s ' ' | tr '\n' ';'
  echo
  sleep 0.5
done > gpurun_out/clk_$label.txt
wait $pid
python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/clk_$label.json').read().strip().splitlines()[-1]); print('$label $wl launch_ms', round(d['roofline']['launch_ms'],4))
except Exception as e: print('$label failed', e)
PY
cat gpurun_out/clk_$label.txt
