#!/bin/bash
# tools/pmc_one.sh <label> <workload> [env...] — one SQ counter pass (kernel-trace + --pmc only) of one workload under a hard timeout;
# per-kernel averages appended to gpurun_out/pmc_one.txt
export TMPDIR=/tmp
ROOT=$(pwd); LABEL=$1; WL=$2; shift 2
OUT=$ROOT/gpurun_out/pmc1_$LABEL; rm -rf $OUT; mkdir -p $OUT
(cd /tmp && env "$@" timeout -k 5 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY \
   -d $OUT/pmc_sq1 -o pmc -- python $ROOT/bench.py --workload $WL --also none --no-cpu-baseline --no-verify --steps 3 --warmup 1 --launches-per-step 1 > $OUT/log.txt 2>&1)
echo "#### $LABEL ($* -- $WL)" >> $ROOT/gpurun_out/pmc_one.txt
timeout 60 python profiles/summarize.py $OUT 2>&1 | grep -v "rocclr\|canvas_clear\|^==" | cut -c1-45,62-140 >> $ROOT/gpurun_out/pmc_one.txt
