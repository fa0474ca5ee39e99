"""tools/fuzz_yuv_stream.py [first [count]] — tests/test_gpu_yuvstream.py::test_random_yuv_stream_ticks (float and integer-matrix RGB layers) and
::test_random_lone_yuv_stream_ticks over many more seeds (run on the GPU box); every eligible launch forced through tick_yuv_stream."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import test_gpu_yuvstream as T
from swiftvideo_amd import compute as sv
from swiftvideo_amd import chipvideo

first = int(sys.argv[1]) if len(sys.argv) > 1 else 200
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
ctx = sv.makeComputeContext(forType="GPU")
chipvideo.set_switch("CHV_YUV_STREAM", "force")
bad = 0
for seed in range(first, first + count):
    for what, fn, arg in (("batch", T.test_random_yuv_stream_ticks, seed), ("batch-int", T.test_random_yuv_stream_ticks, f"int{seed}"),
                          ("lone", T.test_random_lone_yuv_stream_ticks, seed)):
        try:
            fn(ctx, arg)
        except AssertionError as e:
            bad += 1
            print(what, "seed", seed, "FAILED:", str(e)[:300])
print(f"{count} seeds from {first} x (batch, batch-int, lone): {bad} failures")
