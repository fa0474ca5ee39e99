"""tools/fuzz_lanczos.py [first [count]] — tests/test_gpu_parity.py::test_lanczos_random_geometries over many more seeds (run on the GPU box)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import pytest
import test_gpu_parity as T
from swiftvideo_amd import compute as sv

first = int(sys.argv[1]) if len(sys.argv) > 1 else 48
count = int(sys.argv[2]) if len(sys.argv) > 2 else 400
ctx = sv.makeComputeContext(forType="GPU")
bad = skipped = 0
for seed in range(first, first + count):
    try:
        T.test_lanczos_random_geometries(ctx, seed)
    except AssertionError as e:
        bad += 1
        print("seed", seed, "FAILED:", str(e)[:300])
    except pytest.skip.Exception:
        skipped += 1
print(f"{count} seeds from {first}: {bad} failures, {skipped} skipped")
