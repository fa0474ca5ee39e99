#!/bin/bash
# tools/pmc_quick.sh — one SQ counter pass (kernel-trace + --pmc only) per line of tools/pmc.list: "<label> <env...> -- <workload>";
# per-kernel averages into gpurun_out/pmc_quick.txt   (PMC_SET="…": another counter set, at most eight SQ counters)
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p gpurun_out
: > gpurun_out/pmc_quick.txt
while IFS= read -r line; do
  [ -z "$line" ] && continue
  label=${line%% *}; rest=${line#* }; envs=${rest%%--*}; wl=${rest#*-- }
  OUT=$ROOT/gpurun_out/pmcq_$label
  rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && env $envs timeout -k 5 240 rocprofv3 --kernel-trace --pmc ${PMC_SET:-SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY} \
     -d $OUT/pmc_sq1 -o pmc -- python $ROOT/bench.py --workload $wl --also none --no-cpu-baseline --no-verify --steps 3 --warmup 1 --launches-per-step 1 > $OUT/log.txt 2>&1)
  echo "#### $label ($envs -- $wl)" >> gpurun_out/pmc_quick.txt
  python profiles/summarize.py $OUT 2>&1 | grep -v rocclr >> gpurun_out/pmc_quick.txt
done < tools/pmc.list
cat gpurun_out/pmc_quick.txt
