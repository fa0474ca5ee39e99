timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/t.txt
tools/gpu_ab.sh
cat gpurun_out/t.txt
