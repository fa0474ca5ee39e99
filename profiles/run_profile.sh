#!/bin/bash
# profiles/run_profile.sh <tag> [bench args...] — run on the GPU box (via gpurun).
# Collects, for the bench's default workload:
#   1. timeout -k 5 240 rocprofv3 --kernel-trace --stats          -> gpurun_out/prof_<tag>/stats
#   2. separate --pmc passes (never combined with tracing domains beyond kernel-trace)
# and leaves CSVs under gpurun_out/prof_<tag>/ for profiles/summarize.py.
set -u
TAG=${1:-run}; shift || true
ARGS=${@:-"--steps 50 --warmup 5 --no-cpu-baseline --no-verify"}
# PROFILE_CMD="python /root/repo/tools/lanczos_probe.py" profiles/run_profile.sh <tag>  profiles another command instead
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
CMD=${PROFILE_CMD:-"python $ROOT/bench.py $ARGS"}
cd /tmp
timeout -k 5 240 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- $CMD > $OUT/stats.log 2>&1
timeout -k 5 240 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $OUT/pmc_sq1 -o pmc -- $CMD > $OUT/pmc_sq1.log 2>&1
timeout -k 5 240 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_sq2 -o pmc -- $CMD > $OUT/pmc_sq2.log 2>&1
timeout -k 5 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout -k 5 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $CMD > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -40
