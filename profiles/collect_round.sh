#!/bin/bash
# profiles/collect_round.sh <rNN> — run on the GPU box (gpurun): the round's default bench line (all workloads, CPU baseline),
# rocprofv3 kernel-trace stats of that same command, and kernel-trace + separate PMC passes of the headline workload alone.
# Everything lands under gpurun_out/<rNN>/; copy what should be judged into profiles/.
set -u
R=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
export TMPDIR=/tmp
# 1. the default line, exactly as the driver runs it
T0=$(date +%s.%N); python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; python -c "import time,sys; print('default command: %.1f s wall' % (time.time() - float(sys.argv[1])))" $T0 > $OUT/bench_default.time
cp bench_detail.json $OUT/bench_default_detail.json
# 1b. every leg (upload / end-to-end / one tick at a time / thread scaling / route regret / clock and power of every workload): --full
python bench.py --gpus 1 --steps 20 --warmup 5 --full --detail-json $OUT/bench_full_detail.json > $OUT/bench_full.json 2> $OUT/bench_full.err
# 2. kernel-trace stats of every workload's kernel (shorter timed regions: the profiler keeps every dispatch)
(cd /tmp && timeout -k 5 240 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_${R}_all/stats -o stats -- python $ROOT/bench.py --no-cpu-baseline --no-verify --no-live-pmc --no-per-tick --no-upload-leg --no-route-regret --no-power-probe --min-seconds 0.3 --min-seconds-other 0.15 --steps 10 --warmup 3 > $OUT/bench_under_rocprof.json 2> /dev/null)
python profiles/summarize.py gpurun_out/prof_${R}_all > $OUT/all_workloads_rocprofv3.txt 2>&1
# 3a. the headline workload alone under kernel-trace, long enough (hundreds of launches) for the average to be comparable with the
#     HIP-event figure of the default line
(cd /tmp && timeout -k 5 240 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_${R}_headline/stats -o stats -- python $ROOT/bench.py --workload pipeline --also none --no-cpu-baseline --no-verify --min-seconds 0.6 --steps 10 --warmup 3 > $OUT/bench_headline_under_rocprof.json 2> /dev/null)
python profiles/summarize.py gpurun_out/prof_${R}_headline > $OUT/headline_rocprofv3_stats.txt 2>&1
# 3b. the headline kernel alone: stats + PMC passes (each counter set in its own run, never combined with tracing beyond kernel-trace)
bash profiles/run_profile.sh ${R}_pipeline --workload pipeline --also none --no-cpu-baseline --no-verify --steps 5 --warmup 2 --launches-per-step 1 > /dev/null 2>&1
python profiles/summarize.py gpurun_out/prof_${R}_pipeline --kernel tick_bgra_stream --name tick_bgra_stream --workload pipeline --frames 256 \
       --pmc-json $OUT/pmc_latest.json --source profiles/${R}_pipeline_rocprofv3.txt > $OUT/pipeline_rocprofv3.txt 2>&1
# 4. the same for the reference's own kernels on their default canvas
bash profiles/run_profile.sh ${R}_mixer --workload mixer_y420p --also none --no-cpu-baseline --no-verify --steps 5 --warmup 2 --launches-per-step 1 > /dev/null 2>&1
python profiles/summarize.py gpurun_out/prof_${R}_mixer --kernel tick_yuv_wave > $OUT/mixer_y420p_rocprofv3.txt 2>&1
# 4b. the encoder-side frame through the 4:2:0 streaming kernel
bash profiles/run_profile.sh ${R}_encode --workload encode_nv12 --also none --no-cpu-baseline --no-verify --steps 5 --warmup 2 --launches-per-step 1 > /dev/null 2>&1
python profiles/summarize.py gpurun_out/prof_${R}_encode --kernel tick_yuv_stream > $OUT/encode_nv12_rocprofv3.txt 2>&1
# 4c. cfg5 (the strip kernel on eight 2160p RGB layers with LDS-DMA staging, then lanczos3_strip2): stats + counters of both kernels
bash profiles/run_profile.sh ${R}_cfg5 --workload cfg5 --also none --no-cpu-baseline --no-verify --steps 5 --warmup 2 --launches-per-step 1 > /dev/null 2>&1
python profiles/summarize.py gpurun_out/prof_${R}_cfg5 --kernel tick_bgra_wave > $OUT/cfg5_rocprofv3.txt 2>&1
# 5. A/B lines: general kernels, wave kernel on the single-purpose workloads
for w in pipeline mixer_y420p cfg2; do CHV_FORCE_GENERAL=1 python bench.py --workload $w --also none --no-cpu-baseline --min-seconds 0.5 > $OUT/bench_${w}_general_kernel.json 2>/dev/null; done
for w in cfg2 cfg3; do CHV_BGRA_PATH=wave python bench.py --workload $w --also none --no-cpu-baseline --min-seconds 0.5 > $OUT/bench_${w}_wave_kernel.json 2>/dev/null; done
CHV_STREAM=0 python bench.py --workload pipeline --also none --no-cpu-baseline --min-seconds 1.0 > $OUT/bench_pipeline_wave_kernel.json 2>/dev/null
CHV_BGRA_PATH=stream python bench.py --workload cfg2 --also none --no-cpu-baseline --min-seconds 0.5 > $OUT/bench_cfg2_stream_kernel.json 2>/dev/null
CHV_STREAM=0 python bench.py --workload pipeline_y420p --also none --no-cpu-baseline --min-seconds 0.5 > $OUT/bench_pipeline_y420p_wave_kernel.json 2>/dev/null
CHV_YUV_STREAM=0 python bench.py --workload encode_nv12 --also y420p_main --no-cpu-baseline --min-seconds 0.5 > $OUT/bench_yuv_wave_kernel.json 2>/dev/null
CHV_YUV_STREAM=force python bench.py --workload encode_nv12 --also y420p_main,mixer_y420p,mixer_nv12 --no-cpu-baseline --min-seconds 0.5 > $OUT/bench_yuv_stream_kernel.json 2>/dev/null
python bench.py --workload mixed --also y420p_main,mixer_nv12,cfg2_y420p,encode_nv12,pipeline_logo --no-cpu-baseline --min-seconds 0.5 > $OUT/bench_more_workloads.json 2>/dev/null
tail -c 400 $OUT/bench_default.json; echo; tail -5 $OUT/pipeline_rocprofv3.txt
