#!/bin/bash
# tools/pmc_traffic.sh <list> — HBM traffic per launch (FETCH_SIZE and WRITE_SIZE, each in its OWN rocprofv3 pass under a hard timeout) for
# every line "<label> <env...> -- <workload>" of the list; per-kernel averages into gpurun_out/pmc_traffic.txt
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p gpurun_out
: > gpurun_out/pmc_traffic.txt
while IFS= read -r line; do
  [ -z "$line" ] && continue
  label=${line%% *}; rest=${line#* }; envs=${rest%%--*}; wl=${rest#*-- }
  OUT=$ROOT/gpurun_out/pmct_$label
  rm -rf $OUT; mkdir -p $OUT
  for c in FETCH_SIZE WRITE_SIZE; do
    d=pmc_fetch; [ $c = WRITE_SIZE ] && d=pmc_write
    (cd /tmp && env $envs timeout -k 5 120 rocprofv3 --kernel-trace --pmc $c -d $OUT/$d -o pmc -- python $ROOT/bench.py --workload $wl --also none --no-cpu-baseline --no-verify --steps 3 --warmup 1 --launches-per-step 1 > $OUT/$d.log 2>&1)
  done
  echo "#### $label ($envs -- $wl)" >> gpurun_out/pmc_traffic.txt
  timeout 60 python profiles/summarize.py $OUT 2>&1 | grep -v "rocclr\|canvas_clear\|^==" | cut -c1-45,62-140 >> gpurun_out/pmc_traffic.txt
done < $1
cat gpurun_out/pmc_traffic.txt
