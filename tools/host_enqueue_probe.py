"""tools/host_enqueue_probe.py — what the HOST spends per lone tick before the launch is on the queue: chv_composite called N times back to back
without waiting (the device falls behind; nothing here waits for it), per tick kind.  The difference to the empty tick is descriptor building,
route selection, launch planning, the geometry store's lookup and the runtime's own launch path for the kernel's argument block.  GPU box."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import ctypes as C
import util, gpuutil as G
from swiftvideo_amd import compute as sv, chipvideo as cv
ctx = sv.makeComputeContext(forType="GPU")
lib = cv.load()
K = sv.defaultComputeKernelFromString
N = 300


ONLY = sys.argv[1] if len(sys.argv) > 1 else None          # run the tick kinds whose label contains this


def probe(label, dst, layers):
    if ONLY and ONLY not in label:
        return
    tdesc = sv._image_desc(dst)
    arr = sv._layer_array(layers) if layers else None
    n = len(layers)
    comp, h = lib.chv_composite, ctx.handle
    ref = C.byref(tdesc)
    for _ in range(20): comp(h, ref, 1, arr, n)
    lib.chv_pass_end(h, 1)
    best = 1e9
    for rep in range(5):
        t = time.perf_counter()
        for _ in range(N): comp(h, ref, 1, arr, n)
        dt = (time.perf_counter() - t) / N * 1e6
        lib.chv_pass_end(h, 1)
        best = min(best, dt)
    print(f"{label:60s} {best:6.2f} us of host time per chv_composite (best of 5 x {N})", flush=True)


bg = G.to_gpu(ctx, "bgra", 1280, 720, util.alloc_image("bgra", 1280, 720))
nv = [G.to_gpu(ctx, "nv12", 1920, 1080, util.alloc_image("nv12", 1920, 1080, seed=2 + i)) for i in range(4)]
four = []
for s4, o in zip(nv, (1.0, 0.75, 0.5, 0.25)):
    f = s4.derive(matrix=sv._unit_quad_to_ndc(), borderMatrix=sv._unit_quad_to_ndc(), opacity=o)
    four.append((sv.ComputeKernel.img_nv12_bgra, f, sv.imageUniformsFor(f, bg), 0))
probe("empty tick (clear only, 720p BGRA)", bg, [])
probe("cfg2 tick (1 NV12 layer -> 720p BGRA, tick_bgra_stream_one)", bg, four[:1])
probe("pipeline tick (4 NV12 layers, tick_bgra_stream_one)", bg, four)
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
    probe(f"mixer tick, 1080p {fmt} canvas, video + 2 overlays (strip kernel)", dst, layers)
    probe(f"one video layer, 1080p {fmt} canvas", dst, layers[:1])
    for sw in ("0",):
        cv.set_switch("CHV_GEOM_CACHE", sw)
        probe(f"mixer tick, 1080p {fmt} canvas, CHV_GEOM_CACHE=0", dst, layers)
        cv.set_switch("CHV_GEOM_CACHE", None)
