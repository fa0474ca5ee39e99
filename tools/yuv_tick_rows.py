"""tools/yuv_tick_rows.py — the lone 4:2:0 mixer tick (video + two BGRA overlays, one chv_composite + wait) through tick_yuv_wave with 8-row and
16-row strips (CHV_WAVE_ROWS) and by the library's own choice: host wall clock per tick and device time between two stream events.  GPU box."""
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
    for _ in range(60):
        lib.chv_event_record(ctx.handle, e0)
        lib.chv_composite(ctx.handle, C.byref(tdesc), 1, arr, len(layers))
        lib.chv_event_record(ctx.handle, e1)
        lib.chv_pass_end(ctx.handle, 1)
        ms = C.c_float(); lib.chv_event_elapsed_ms(e0, e1, C.byref(ms)); dev.append(ms.value * 1e3)
    dev.sort()
    print(f"{label:64s} wall {wall:6.1f} us/tick   device median {dev[len(dev)//2]:6.1f} us  min {dev[0]:6.1f}", flush=True)


K = sv.defaultComputeKernelFromString
for fmt in ("y420p", "nv12", "bgra"):
    dst = G.to_gpu(ctx, fmt, 1920, 1080, util.alloc_image(fmt, 1920, 1080))
    vf = "nv12" if fmt == "bgra" else fmt
    src = G.to_gpu(ctx, vf, 1920, 1080, util.alloc_image(vf, 1920, 1080, seed=9))
    ov = [G.to_gpu(ctx, "bgra", 640, 360, util.alloc_image("bgra", 640, 360, seed=10 + i)) for i in range(2)]
    full = util.full_canvas_uniforms((1920, 1080), (1920, 1080))
    ovk = "img_bgra_bgra_tx" if fmt == "bgra" else f"img_bgra_{fmt}"
    mixer = [(K(f"img_{vf}_{fmt}"), src, full, 0)] + [
        (K(ovk), o, util.make_uniforms((1920, 1080), rect=(px, py, 640, 360), opacity=op, in_size=(640, 360)), 0)
        for o, (px, py), op in zip(ov, ((64, 64), (1200, 640)), (0.8, 0.6))]
    tdesc = sv._image_desc(dst)
    for rows in (None, "8", "16"):
        cv.set_switch("CHV_WAVE_ROWS", rows)
        for _ in range(2):
            probe(f"{fmt} canvas, video + 2 overlays, CHV_WAVE_ROWS={rows}", tdesc, mixer)
    cv.set_switch("CHV_WAVE_ROWS", None)
