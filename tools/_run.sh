timeout 1200 python -m pytest tests/test_gpu_yuvwave.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_mixer.py -x -q 2>&1 | tail -8 > gpurun_out/t.txt
cat gpurun_out/t.txt
tools/pmc_quick.sh > /dev/null 2>&1
grep -E '####|SQ_' gpurun_out/pmc_quick.txt | cut -c1-30,62-130
