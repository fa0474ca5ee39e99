"""tools/lanczos_probe.py [iw ih ow oh [n]] — device time of chv_scale_lanczos_batch for n resizes of one geometry (default 3840x2160 -> 1920x1080, n = 24),
HIP events on the context's stream; prints microseconds per batch."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import util, gpuutil as G
from swiftvideo_amd import compute as sv

a = [int(x) for x in sys.argv[1:]]
iw, ih, ow, oh = (a + [3840, 2160, 1920, 1080])[:4] if len(a) >= 4 else (3840, 2160, 1920, 1080)
n = a[4] if len(a) > 4 else 24
ctx = sv.makeComputeContext(forType="GPU")
srcs = [util.alloc_image("bgra", iw, ih, seed=1 + i) for i in range(min(n, 3))]
pairs = [(G.to_gpu(ctx, "bgra", ow, oh, util.alloc_image("bgra", ow, oh)), G.to_gpu(ctx, "bgra", iw, ih, srcs[i % len(srcs)])) for i in range(n)]
batch = sv.LanczosBatch(pairs)
for _ in range(5):
    sv.usingContext(ctx, lambda c: batch.run(c))
reps = 50
t = time.perf_counter()
sv.usingContext(ctx, lambda c: [batch.run(c) for _ in range(reps)] and c)
dt = (time.perf_counter() - t) / reps
print(f"{iw}x{ih} -> {ow}x{oh} x {n}: {dt * 1e6:.1f} us per batch, {dt / n * 1e6:.2f} us per image, {n * (iw * ih + ow * oh) * 4 / dt / 1e9:.0f} GB/s algorithmic")
