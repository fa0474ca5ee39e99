timeout 1500 python -m pytest tests/test_gpu_yuvwave.py tests/test_gpu_parity.py tests/test_gpu_mixer.py -x -q 2>&1 | tail -4 > gpurun_out/t.txt
tools/gpu_ab.sh
cat gpurun_out/t.txt
