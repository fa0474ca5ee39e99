#!/bin/bash
# profiles/collect_round.sh <rNN> — run on the GPU box (gpurun): the round's bench lines (with CPU baseline),
# the rocprofv3 kernel-trace + PMC passes of the default workload, kernel-trace stats of the other workloads.
# Everything lands under gpurun_out/<rNN>/; copy what should be judged into profiles/.
set -u
R=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
python bench.py > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err
python bench.py --workload cfg3 > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err
python bench.py --workload cfg5 > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err
python bench.py --workload mixer_y420p --no-cpu-baseline > $OUT/bench_mixer_y420p.json 2> $OUT/bench_mixer.err
python bench.py --workload cfg2_y420p --no-cpu-baseline > $OUT/bench_cfg2_y420p.json 2> $OUT/bench_cfg2_y420p.err
CHV_FORCE_GENERAL=1 python bench.py --no-cpu-baseline > $OUT/bench_cfg2_general_kernel.json 2>/dev/null
CHV_FORCE_GENERAL=1 python bench.py --workload cfg3 --no-cpu-baseline > $OUT/bench_cfg3_general_kernel.json 2>/dev/null
bash profiles/run_profile.sh ${R}_cfg2 > /dev/null 2>&1
python profiles/summarize.py gpurun_out/prof_${R}_cfg2 --pmc-json $OUT/pmc_latest.json --source profiles/${R}_cfg2_rocprofv3.txt > $OUT/cfg2_rocprofv3.txt 2>&1
export TMPDIR=/tmp
for w in cfg3 cfg5; do
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_${R}_$w/stats -o stats -- python $ROOT/bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-verify > /dev/null 2>&1)
  python profiles/summarize.py gpurun_out/prof_${R}_$w > $OUT/${w}_rocprofv3.txt 2>&1
done
tail -c 600 $OUT/bench_cfg2.json; echo; cat $OUT/cfg2_rocprofv3.txt | tail -5
