#!/bin/bash
# tools/cfg5_groups.sh — cfg5 (8 x 2160p layers -> 2160p canvas -> Lanczos-3 -> 1080p) with the two stages issued in groups of G ticks
mkdir -p gpurun_out
for g in 24 8 4 2 1; do
  timeout 200 python bench.py --workload cfg5 --also none --no-cpu-baseline --min-seconds 0.6 --steps 10 --warmup 3 --group $g 2>/dev/null | tail -1 | \
  python -c "import sys,json
d=json.loads(sys.stdin.read()); w=json.load(open('bench_detail.json'))['workloads']['cfg5']; print('group $g', 'launch_ms', round(d['roofline']['launch_ms'],4), 'frac', round(d['roofline']['frac'],4), 'launches', w.get('kernel_launches_per_batch'), 'verified', d['config']['verified_vs_oracle'])"
done | tee gpurun_out/cfg5_groups.txt
