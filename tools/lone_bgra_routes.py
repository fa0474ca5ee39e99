"""tools/lone_bgra_routes.py — a lone tick of ONE video layer on a cleared BGRA canvas through each of its three routes (streaming twin — the
default below 1.4 Mpixel —, tiled twin, strip twin): 720p and 1080p canvases, NV12 and y420p sources, full canvas and a picture-in-picture inset;
wall us per tick with the host wait and us between two stream events.  GPU box.  (profiles/r06_notes.md section 18.)"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import ctypes as C
import util, gpuutil as G
from swiftvideo_amd import compute as sv, chipvideo as cv
ctx = sv.makeComputeContext(forType="GPU")
lib = cv.load()
e0, e1 = C.c_void_p(), C.c_void_p()
cv.check(lib.chv_event_create(ctx.handle, C.byref(e0))); cv.check(lib.chv_event_create(ctx.handle, C.byref(e1)))
def probe(label, tdesc, layers, n=400):
    arr = sv._layer_array(layers)
    def tick():
        cv.check(lib.chv_composite(ctx.handle, C.byref(tdesc), 1, arr, len(layers))); lib.chv_pass_end(ctx.handle, 1)
    for _ in range(50): tick()
    t = time.perf_counter()
    for _ in range(n): tick()
    wall = (time.perf_counter() - t) / n * 1e6
    dev = []
    for _ in range(50):
        lib.chv_event_record(ctx.handle, e0); lib.chv_composite(ctx.handle, C.byref(tdesc), 1, arr, len(layers)); lib.chv_event_record(ctx.handle, e1); lib.chv_pass_end(ctx.handle, 1)
        ms = C.c_float(); lib.chv_event_elapsed_ms(e0, e1, C.byref(ms)); dev.append(ms.value * 1e3)
    dev.sort()
    print(f"{label:70s} wall {wall:6.1f} us   device {dev[len(dev)//2]:6.1f}", flush=True)
K = sv.defaultComputeKernelFromString
for (cw, ch) in ((1280, 720), (1920, 1080)):
    dst = G.to_gpu(ctx, "bgra", cw, ch, util.alloc_image("bgra", cw, ch))
    td = sv._image_desc(dst)
    for vf in ("nv12", "y420p"):
        src = G.to_gpu(ctx, vf, 1920, 1080, util.alloc_image(vf, 1920, 1080, seed=3))
        full = util.full_canvas_uniforms((cw, ch), (1920, 1080))
        inset = util.make_uniforms((cw, ch), rect=(cw // 8, ch // 8, cw // 2, ch // 2), in_size=(1920, 1080))
        for gname, u in (("full canvas", full), ("inset", inset)):
            for route in (None, "tiled", "wave"):
                cv.set_switch("CHV_BGRA_PATH", route)
                probe(f"{cw}x{ch} bgra <- {vf} {gname}, CHV_BGRA_PATH={route}", td, [(K(f"img_{vf}_bgra"), src, u, 0)])
            cv.set_switch("CHV_BGRA_PATH", None)
