#!/bin/bash
# tools/gpu_iter2.sh — full default bench line + wave path on the single-purpose workloads + 2-rank self-launch on one device
mkdir -p gpurun_out
{
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
echo "== wave path on cfg2 / cfg3 / cfg2_y420p (vs single-purpose kernels)"
for mode in single wave; do for wl in cfg2 cfg3 cfg2_y420p; do
  CHV_BGRA_PATH=$mode timeout 600 python bench.py --workload $wl --also none --no-cpu-baseline --min-seconds 0.5 --steps 10 --warmup 3 2>&1 | tail -1 | \
    python -c "import sys,json
try:
    d=json.loads(sys.stdin.read()); print('$mode $wl', d['config']['kernel'], 'launch_ms', round(d['roofline']['launch_ms'],4), 'frac', round(d['roofline']['frac'],4), 'verified', d['config']['verified_vs_oracle'])
except Exception as e: print('$mode $wl FAILED', e)"
done; done
echo "== full default run"; ( time timeout 900 python bench.py ) > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err; tail -3 gpurun_out/bench_full.err
python -c "import json; d=json.loads(open('gpurun_out/bench_full.json').read().strip().splitlines()[-1]); print('headline', d['value'], d['roofline']['frac'], d['config']['launches_per_step'], d['config']['timed_seconds']); [print(k, round(v['launch_ms'],4) if 'launch_ms' in v else '', v.get('roofline',{}).get('frac'), v.get('verified_vs_oracle'), v['kernel'], v.get('h2d_GBps_per_gpu')) for k,v in d['workloads'].items()]; print(d.get('cpu_baseline'))"
echo "== --gpus 2 --device 0 (self-launch)"; timeout 600 python bench.py --gpus 2 --device 0 --also none --no-cpu-baseline --min-seconds 0.5 --steps 5 --warmup 2 --frames 64 2>&1 | tail -2 | cut -c1-600
} > gpurun_out/iter2.txt 2>&1
cat gpurun_out/iter2.txt
