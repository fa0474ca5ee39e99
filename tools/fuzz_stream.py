"""tools/fuzz_stream.py [first [count]] — tests/test_gpu_mixpath.py::test_random_stream_ticks and ::test_random_lone_stream_ticks over many
more seeds (run on the GPU box)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import test_gpu_mixpath as T
from swiftvideo_amd import compute as sv

first = int(sys.argv[1]) if len(sys.argv) > 1 else 24
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
from swiftvideo_amd import chipvideo
ctx = sv.makeComputeContext(forType="GPU")
bad, streamed = 0, 0
names = []
_orig = T.G.make_batch
def _mb(*a, **k):
    r = _orig(*a, **k); names.append(r[1]); return r
T.G.make_batch = _mb
for seed in range(first, first + count):
    try:
        T.test_random_stream_ticks(ctx, lambda n, v: chipvideo.set_switch(n, v), seed)
        chipvideo.set_switch("CHV_BGRA_PATH", None)
        T.test_random_lone_stream_ticks(ctx, seed)        # the same ticks one at a time (descriptors as kernel arguments)
    except AssertionError as e:
        bad += 1
        print("seed", seed, "FAILED:", str(e)[:300])
print(f"{count} seeds from {first}: {bad} failures; kernels: " + ", ".join(f"{n} x {names.count(n)}" for n in sorted(set(names))))
