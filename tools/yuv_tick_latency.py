"""tools/yuv_tick_latency.py — ONE 4:2:0 tick at a time (chv_composite + the reference's wait), through tick_yuv_stream (descriptors as kernel
arguments, short chunks) and through tick_yuv_wave (CHV_YUV_STREAM=0): host wall clock per tick and device time between two stream events.
Run on the GPU box."""
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


def probe(label, tdesc, layers, n=300):
    arr = sv._layer_array(layers)
    def tick():
        cv.check(lib.chv_composite(ctx.handle, C.byref(tdesc), 1, arr, len(layers))); lib.chv_pass_end(ctx.handle, 1)
    for _ in range(50): tick()
    t = time.perf_counter()
    for _ in range(n): tick()
    wall = (time.perf_counter() - t) / n * 1e6
    dev = []
    for _ in range(50):
        lib.chv_event_record(ctx.handle, e0)
        lib.chv_composite(ctx.handle, C.byref(tdesc), 1, arr, len(layers))
        lib.chv_event_record(ctx.handle, e1)
        lib.chv_pass_end(ctx.handle, 1)
        ms = C.c_float(); lib.chv_event_elapsed_ms(e0, e1, C.byref(ms)); dev.append(ms.value * 1e3)
    dev.sort()
    print(f"{label:52s} wall {wall:6.1f} us/tick   device (events) median {dev[len(dev)//2]:6.1f} us  min {dev[0]:6.1f}", flush=True)


W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)          # canvas (and video) size; the overlays are a third of it


def ticks(fmt):
    dst = G.to_gpu(ctx, fmt, W, H, util.alloc_image(fmt, W, H))
    src = G.to_gpu(ctx, fmt, W, H, util.alloc_image(fmt, W, H, seed=9))
    ow, oh = W // 3 // 2 * 2, H // 3 // 2 * 2
    ov = [G.to_gpu(ctx, "bgra", ow, oh, util.alloc_image("bgra", ow, oh, seed=10 + i)) for i in range(2)]
    rgb = G.to_gpu(ctx, "bgra", W, H, util.alloc_image("bgra", W, H, seed=12))
    full = util.full_canvas_uniforms((W, H), (W, H))
    K = sv.defaultComputeKernelFromString
    main = [(K(f"img_{fmt}_{fmt}"), src, full, 0)]
    mixer = main + [(K(f"img_bgra_{fmt}"), o, util.make_uniforms((W, H), rect=(px, py, ow, oh), opacity=op, in_size=(ow, oh)), 0)
                    for o, (px, py), op in zip(ov, ((W // 30, H // 17), (W * 5 // 8, H * 16 // 27)), (0.8, 0.6))]
    enc = [(K(f"img_bgra_{fmt}_int"), rgb, full, 0)]
    return sv._image_desc(dst), main, mixer, enc, (dst, src, ov, rgb)


for fmt in ("y420p", "nv12"):
    tdesc, main, mixer, enc, keep = ticks(fmt)
    for sw in ("1", "0"):
        cv.set_switch("CHV_YUV_STREAM", sw)
        tag = "tick_yuv_stream" if sw == "1" else "tick_yuv_wave"
        probe(f"{W}x{H} {fmt} {tag}: one video layer", tdesc, main)
        probe(f"{fmt} {tag}: mixer tick (video + 2 overlays)", tdesc, mixer)
        probe(f"{fmt} {tag}: encoder frame (BGRA -> {fmt}, int)", tdesc, enc)
    cv.set_switch("CHV_YUV_STREAM", None)
