bash tools/pmc_quick.sh > /dev/null 2>&1
export TMPDIR=/tmp
ROOT=$(pwd)
for v in 1 0; do
OUT=$ROOT/gpurun_out/pmct_$v; rm -rf $OUT; mkdir -p $OUT
(cd /tmp && CHV_YUV_STREAM=$v rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM -d $OUT/p -o pmc -- python $ROOT/bench.py --workload y420p_main --also none --no-cpu-baseline --no-verify --steps 3 --warmup 1 --launches-per-step 1 > $OUT/log.txt 2>&1)
echo "#### traffic CHV_YUV_STREAM=$v" >> gpurun_out/pmc_quick.txt
python profiles/summarize.py $OUT 2>&1 | grep -v rocclr >> gpurun_out/pmc_quick.txt
done
grep -v "canvas_clear\|^==" gpurun_out/pmc_quick.txt | cut -c1-40,62-140
