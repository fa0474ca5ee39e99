#!/bin/bash
# tools/power_classes.sh — for each vector instruction class: sustained issue time, shader clock and socket power over ~3 s (GPU box);
# (the readings printed ABOVE a class's line belong to it) gpurun_out/power_classes.txt.  Build first: hipcc --offload-arch=gfx950 -O2 tools/ubench_lds.cpp -o tools/ubench_lds.bin and hipcc --offload-arch=gfx950 -O2 tools/ubench_power.cpp -o tools/ubench_power.bin
mkdir -p gpurun_out
{
for op in ${VALU_CLASSES:-0 1 2 3 4 8 10 11 12 14 21 23 26 13 31 40}; do
  tools/ubench_power.bin $op 3.0 &
  pid=$!
  sleep 1.6
  for i in 1 2; do rocm-smi --showclocks --showpower --json 2>/dev/null | grep '^{' | python3 -c "
import sys,json
for l in sys.stdin:
    c=next(iter(json.loads(l).values())); print('   sclk', c.get('sclk clock speed:'), 'power', c.get('Current Socket Graphics Package Power (W)'))"; sleep 0.4; done
  wait $pid
done
if [ -x tools/ubench_lds.bin ]; then
for op in 0 1 2 3 4 5 6 7; do
  tools/ubench_lds.bin $op 3.0 &
  pid=$!
  sleep 1.6
  for i in 1 2; do rocm-smi --showclocks --showpower --json 2>/dev/null | grep '^{' | python3 -c "
import sys,json
for l in sys.stdin:
    c=next(iter(json.loads(l).values())); print('   sclk', c.get('sclk clock speed:'), 'power', c.get('Current Socket Graphics Package Power (W)'))"; sleep 0.4; done
  wait $pid
done
fi
} > gpurun_out/power_classes.txt 2>&1
cat gpurun_out/power_classes.txt
