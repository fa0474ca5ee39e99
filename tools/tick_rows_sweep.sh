# tools/tick_rows_sweep.sh — one tick at a time (chv_composite + host wait), 1 / 2 / 4 NV12 layers: default routing, the strip kernel
# (CHV_STREAM=0), the streaming kernel on request (one-layer ticks too).  GPU box; writes gpurun_out/tick_rows.txt
mkdir -p gpurun_out
{
LABEL=default timeout 120 python tools/tick_rows_probe.py 2>&1 | tail -3
LABEL=strip_kernel CHV_STREAM=0 timeout 120 python tools/tick_rows_probe.py 2>&1 | tail -3
LABEL=stream_forced CHV_BGRA_PATH=stream timeout 120 python tools/tick_rows_probe.py 2>&1 | tail -3
LABEL=default_again timeout 120 python tools/tick_rows_probe.py 2>&1 | tail -3
} > gpurun_out/tick_rows.txt 2>&1
cat gpurun_out/tick_rows.txt
