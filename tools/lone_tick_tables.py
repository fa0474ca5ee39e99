"""tools/lone_tick_tables.py — would a LONE tick gain from geometry tables?  The reference-default mixer tick (1080p y420p canvas <- y420p video + two
BGRA overlays) and the mixed BGRA tick as ONE-tick batches run + waited for one at a time, with the batch's tables (CHV_GEOM_CACHE default) and
with the geometry computed in place (=0): wall us per tick and device us between two events; then the same tick through chv_composite (a transient
launch: what an unmodified VideoMixer issues), which since the device's table store (csrc/geom_cache.h) finds the scene's tables from its third
sighting on — CHV_GEOM_CACHE default against =0 in turn.  GPU box."""
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
K = sv.defaultComputeKernelFromString


def probe(label, h, n=400):
    def tick():
        G.run_batch(ctx, h); lib.chv_pass_end(ctx.handle, 1)
    for _ in range(50): tick()
    t = time.perf_counter()
    for _ in range(n): tick()
    wall = (time.perf_counter() - t) / n * 1e6
    dev = []
    for _ in range(60):
        lib.chv_event_record(ctx.handle, e0); G.run_batch(ctx, h); lib.chv_event_record(ctx.handle, e1); lib.chv_pass_end(ctx.handle, 1)
        ms = C.c_float(); lib.chv_event_elapsed_ms(e0, e1, C.byref(ms)); dev.append(ms.value * 1e3)
    dev.sort()
    print(f"{label:60s} wall {wall:6.1f} us/tick   device median {dev[len(dev)//2]:6.1f} us  min {dev[0]:6.1f}", flush=True)


for fmt in ("y420p", "bgra"):
    dst = G.to_gpu(ctx, fmt, 1920, 1080, util.alloc_image(fmt, 1920, 1080))
    vf = "nv12" if fmt == "bgra" else fmt
    src = G.to_gpu(ctx, vf, 1920, 1080, util.alloc_image(vf, 1920, 1080, seed=9))
    ov = [G.to_gpu(ctx, "bgra", 640, 360, util.alloc_image("bgra", 640, 360, seed=10 + i)) for i in range(2)]
    full = util.full_canvas_uniforms((1920, 1080), (1920, 1080))
    ovk = "img_bgra_bgra_tx" if fmt == "bgra" else f"img_bgra_{fmt}"
    layers = [(K(f"img_{vf}_{fmt}"), src, full, 0)] + [
        (K(ovk), o, util.make_uniforms((1920, 1080), rect=(px, py, 640, 360), opacity=op, in_size=(640, 360)), 0)
        for o, (px, py), op in zip(ov, ((64, 64), (1200, 640)), (0.8, 0.6))]
    for sw in (None, "0", None, "0"):
        cv.set_switch("CHV_GEOM_CACHE", sw)
        h, name, keep = G.make_batch(ctx, [(dst, True, layers)])
        probe(f"{fmt} canvas, video + 2 overlays, {name}, CHV_GEOM_CACHE={sw}", h)
        G.destroy_batch(h)
    tdesc, arr = sv._image_desc(dst), sv._layer_array(layers)

    class Lone:            # (probe() runs a "batch" through G.run_batch: the same bracket around chv_composite)
        pass
    run_batch = G.run_batch
    G.run_batch = lambda c, h: (lib.chv_pass_begin(c.handle), lib.chv_composite(c.handle, C.byref(tdesc), 1, arr, len(layers)), lib.chv_pass_end(c.handle, 1))
    try:
        for sw in (None, "0", None, "0"):
            cv.set_switch("CHV_GEOM_CACHE", sw)
            p0 = cv.get_counter("geom_store_patched")
            probe(f"{fmt} canvas, video + 2 overlays, chv_composite, CHV_GEOM_CACHE={sw}", None)
            print(f"    (launches pointed at the store's tables: {cv.get_counter('geom_store_patched') - p0})")
    finally:
        G.run_batch = run_batch
    cv.set_switch("CHV_GEOM_CACHE", None)
