import sys, time
from pathlib import Path
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import ctypes as C, util, gpuutil as G
from swiftvideo_amd import compute as sv, chipvideo as cv
ctx = sv.makeComputeContext(forType="GPU")
src = G.to_gpu(ctx, "bgra", 3840, 2160, util.alloc_image("bgra", 3840, 2160, seed=1))
dst = G.to_gpu(ctx, "bgra", 1920, 1080, util.alloc_image("bgra", 1920, 1080))
lib = cv.load(); d, s = sv._image_desc(dst), sv._image_desc(src)
for _ in range(20): lib.chv_scale_lanczos(ctx.handle, C.byref(d), C.byref(s))
lib.chv_pass_end(ctx.handle, 1)
t=time.perf_counter()
for _ in range(500): lib.chv_scale_lanczos(ctx.handle, C.byref(d), C.byref(s))
lib.chv_pass_end(ctx.handle, 1)
print(f"{(time.perf_counter()-t)/500*1e6:.1f} us per 2160p->1080p lanczos")
