"""tools/fuzz_wave.py [first [count]] — the strip kernels over many more seeds than the suite runs: tests/test_gpu_mixpath.py::test_random_mixed_ticks
(BGRA canvases) and tests/test_gpu_yuvwave.py::test_random_yuv_ticks (4:2:0 canvases), each with 8- and 16-row strips (run on the GPU box).  The
random geometry includes strong reductions (the pair form of tall rectangles), flips, borders, fill and layers across the canvas edges."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import test_gpu_mixpath as M
import test_gpu_yuvwave as Y
from swiftvideo_amd import compute as sv
from swiftvideo_amd import chipvideo

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 150
ctx = sv.makeComputeContext(forType="GPU")
chipvideo.set_switch("CHV_BGRA_PATH", "wave")
chipvideo.set_switch("CHV_YUV_STREAM", "0")
bad = 0
for rows in ("8", "16"):
    chipvideo.set_switch("CHV_WAVE_ROWS", rows)
    for seed in range(first, first + count):
        for what, fn in (("bgra", M.test_random_mixed_ticks), ("yuv", Y.test_random_yuv_ticks)):
            try:
                fn(ctx, rows, seed)
            except AssertionError as e:
                bad += 1
                print(what, "rows", rows, "seed", seed, "FAILED:", str(e)[:300])
print(f"{count} seeds from {first} x (bgra, yuv) x (8, 16 rows): {bad} failures")
