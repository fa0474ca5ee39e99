# tools/stream_threshold_sweep.sh — ticks per launch at which tick_bgra_stream overtakes the strip kernel (GPU box): writes gpurun_out/tick_sweep.txt
mkdir -p gpurun_out
{
for n in 2 3 4 6 8 12 16 32 64; do
 for mode in wave stream; do
  CHV_BGRA_PATH=$mode timeout 300 python bench.py --workload pipeline --frames $n --also none --no-cpu-baseline --no-live-pmc --min-seconds 0.4 --steps 10 --warmup 3 2>&1 | tail -1 | python -c "import sys,json
d=json.loads(sys.stdin.read()); print('ticks $n $mode', d['config']['kernel'], 'launch_us', round(d['roofline']['launch_ms']*1000,2), 'us_per_tick', round(d['roofline']['launch_ms']*1000/$n,3))"
 done
done
} > gpurun_out/tick_sweep.txt 2>&1
cat gpurun_out/tick_sweep.txt
