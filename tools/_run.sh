export TMPDIR=/tmp
ROOT=$(pwd); R=r02; OUT=$ROOT/gpurun_out/$R; mkdir -p $OUT
(cd /tmp && rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_${R}_headline/stats -o stats -- python $ROOT/bench.py --workload pipeline --also none --no-cpu-baseline --no-verify --min-seconds 0.6 --steps 10 --warmup 3 > $OUT/bench_headline_under_rocprof.json 2> /dev/null)
python profiles/summarize.py gpurun_out/prof_${R}_headline > $OUT/headline_rocprofv3_stats.txt 2>&1
cat $OUT/headline_rocprofv3_stats.txt; tail -c 600 $OUT/bench_headline_under_rocprof.json
