#!/bin/bash
# tools/pmc_deep.sh <workload> — latency / stall counters of one workload's kernel, three separate SQ passes (kernel-trace + --pmc only)
export TMPDIR=/tmp
ROOT=$(pwd); WL=${1:-pipeline}
OUT=$ROOT/gpurun_out/pmcdeep_$WL; rm -rf $OUT; mkdir -p $OUT
CMD="python $ROOT/bench.py --workload $WL --also none --no-cpu-baseline --no-verify --steps 3 --warmup 1 --launches-per-step 1"
cd /tmp
timeout -k 5 240 rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS -d $OUT/a -o pmc -- $CMD > $OUT/a.log 2>&1
timeout -k 5 240 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC -d $OUT/b -o pmc -- $CMD > $OUT/b.log 2>&1
timeout -k 5 240 rocprofv3 --kernel-trace --pmc SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES -d $OUT/c -o pmc -- $CMD > $OUT/c.log 2>&1
cd $ROOT
python profiles/summarize.py $OUT 2>&1 | grep -v rocclr > gpurun_out/pmc_deep_$WL.txt
cat gpurun_out/pmc_deep_$WL.txt | cut -c1-30,62-130
