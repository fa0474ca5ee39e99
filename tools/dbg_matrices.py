import sys, ctypes as C
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
from swiftvideo_amd import compute as sv, chipvideo as cv
from test_rgb_to_yuv_int import rgb2yuv
ctx = sv.makeComputeContext(forType="GPU")
lib = cv.load()
fn = lib.chv_selftest_matrices; fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
out = np.empty(1 << 24, dtype=np.uint32); m = C.c_uint32(0)
cv.check(fn(ctx.handle, 1, 3, out.ctypes.data, C.byref(m)))
for (r, g, b) in [(0, 0, 7), (0, 0, 100), (0, 0, 255), (7, 0, 0), (255, 0, 0), (0, 7, 0), (0, 255, 0), (255, 255, 255), (17, 99, 203)]:
    w = int(out[(r << 16) | (g << 8) | b])
    print((r, g, b), "device", (w & 255, (w >> 8) & 255, (w >> 16) & 255, w >> 24), "formula", rgb2yuv(3, r, g, b))
