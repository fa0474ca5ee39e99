# tools/stream_rows_sweep.sh — rows per chunk of tick_bgra_stream against ticks per launch (GPU box): gpurun_out/rows_sweep.txt
# Build the variants first (CPU container):  for r in 4 6 8 12 16 24; do bash tools/build_variant.sh rows$r kernels_stream.hip.cpp -DCHV_STREAM_ROWS_FIXED=$r; done
mkdir -p gpurun_out
{
for n in 1 2 4 8 16 32; do
 for r in 4 6 8 12 16 24; do
  CHV_LIB=variants/rows$r.so CHV_BGRA_PATH=stream timeout 300 python bench.py --workload pipeline --frames $n --also none --no-cpu-baseline --no-live-pmc --min-seconds 0.3 --steps 10 --warmup 3 2>&1 | tail -1 | python -c "import sys,json
d=json.loads(sys.stdin.read()); print('ticks $n rows $r', d['config']['kernel'], 'launch_us', round(d['roofline']['launch_ms']*1000,2))"
 done
done
} > gpurun_out/rows_sweep.txt 2>&1
cat gpurun_out/rows_sweep.txt
