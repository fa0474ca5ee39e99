#!/bin/bash
# tools/runtime_wait_knobs.sh — the lone tick (launch + hipStreamSynchronize) under the ROCm runtime's wait / dispatch environment knobs; GPU box.
# Each line: the knob, then tools/tick_latency.py's lines (wall us per tick, device us between two events).
mkdir -p gpurun_out
{
for knobs in "X=1" "HSA_ENABLE_INTERRUPT=0" "ROC_ACTIVE_WAIT_TIMEOUT=1000" "HSA_ENABLE_INTERRUPT=0 ROC_ACTIVE_WAIT_TIMEOUT=1000" "AMD_DIRECT_DISPATCH=0" "HIP_FORCE_DEV_KERNARG=0" "HIP_FORCE_DEV_KERNARG=1" "GPU_MAX_HW_QUEUES=1" "X=2"; do
  echo "== $knobs"
  env $knobs timeout 300 python tools/tick_latency.py 2>&1 | sed -n 2,8p
done
} > gpurun_out/runtime_wait_knobs.txt 2>&1
cat gpurun_out/runtime_wait_knobs.txt
