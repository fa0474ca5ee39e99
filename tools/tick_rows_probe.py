"""tools/tick_rows_probe.py — one 4-layer pipeline tick at a time (chv_composite + wait) through the kernel named by CHV_BGRA_PATH (env): wall µs
per tick and device µs between two events.  Run on the GPU box."""
import sys, time, os
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import ctypes as C
import util, gpuutil as G
from swiftvideo_amd import compute as sv, chipvideo as cv
ctx = sv.makeComputeContext(forType="GPU")
lib = cv.load()
dst = G.to_gpu(ctx, "bgra", 1280, 720, util.alloc_image("bgra", 1280, 720))
tdesc = sv._image_desc(dst)
srcs = [G.to_gpu(ctx, "nv12", 1920, 1080, util.alloc_image("nv12", 1920, 1080, seed=2 + i)) for i in range(4)]
four = []
for s4, o in zip(srcs, (1.0, 0.75, 0.5, 0.25)):
    f = s4.derive(matrix=sv._unit_quad_to_ndc(), borderMatrix=sv._unit_quad_to_ndc(), opacity=o)
    four.append((sv.ComputeKernel.img_nv12_bgra, f, sv.imageUniformsFor(f, dst), 0))
e0, e1 = C.c_void_p(), C.c_void_p()
cv.check(lib.chv_event_create(ctx.handle, C.byref(e0))); cv.check(lib.chv_event_create(ctx.handle, C.byref(e1)))
for nl in (1, 2, 4):
    arr = sv._layer_array(four[:nl])
    def tick():
        lib.chv_composite(ctx.handle, C.byref(tdesc), 1, arr, nl); lib.chv_pass_end(ctx.handle, 1)
    for _ in range(100): tick()
    n = 600
    t = time.perf_counter()
    for _ in range(n): tick()
    wall = (time.perf_counter() - t) / n * 1e6
    dev = []
    for _ in range(100):
        lib.chv_event_record(ctx.handle, e0); lib.chv_composite(ctx.handle, C.byref(tdesc), 1, arr, nl); lib.chv_event_record(ctx.handle, e1)
        lib.chv_pass_end(ctx.handle, 1)
        ms = C.c_float(); lib.chv_event_elapsed_ms(e0, e1, C.byref(ms)); dev.append(ms.value * 1e3)
    dev.sort()
    name = C.c_char_p(lib.chv_last_kernel(ctx.handle)).value if hasattr(lib, "chv_last_kernel") else b"?"
    print(f"{os.environ.get('LABEL','')} layers {nl} kernel {name.decode()} wall {wall:6.1f} us/tick  device median {dev[len(dev)//2]:6.1f} min {dev[0]:6.1f}", flush=True)
