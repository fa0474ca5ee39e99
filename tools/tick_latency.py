"""tools/tick_latency.py — where the time of ONE mixer tick goes (chv_composite + the reference's wait): host wall clock per tick and
device time between two stream events around the launch, for an empty tick (clear only: launch + wait overhead), the cfg2 tick and
the 4-layer headline tick, with the strip heights the host can pick.  Run on the GPU box."""
import sys, time
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

def probe(label, layers, n=400):
    arr = sv._layer_array(layers) if layers else None
    def tick():
        lib.chv_composite(ctx.handle, C.byref(tdesc), 1, arr, len(layers)); lib.chv_pass_end(ctx.handle, 1)
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
    print(f"{label:42s} wall {wall:6.1f} us/tick   device (events) median {dev[len(dev)//2]:6.1f} us  min {dev[0]:6.1f}", flush=True)

ydst = G.to_gpu(ctx, "y420p", 1920, 1080, util.alloc_image("y420p", 1920, 1080))
ysrc = G.to_gpu(ctx, "y420p", 1920, 1080, util.alloc_image("y420p", 1920, 1080, seed=9))
yov = [G.to_gpu(ctx, "bgra", 640, 360, util.alloc_image("bgra", 640, 360, seed=10 + i)) for i in range(2)]
ydesc = sv._image_desc(ydst)


def probe_yuv(n=300):
    """the reference-default mixer tick: 1080p y420p canvas <- 1080p y420p layer + two 640x360 BGRA overlays"""
    global tdesc
    f0 = ysrc.derive(matrix=sv._unit_quad_to_ndc(), borderMatrix=sv._unit_quad_to_ndc())
    layers = [(sv.ComputeKernel.img_y420p_y420p, f0, sv.imageUniformsFor(f0, ydst), 0)]
    for o, (px, py) in zip(yov, ((64, 64), (1200, 640))):
        layers.append((sv.ComputeKernel.img_bgra_y420p, o, util.make_uniforms((1920, 1080), rect=(px, py, 640, 360), opacity=0.8, in_size=(640, 360)), 0))
    keep, tdesc = tdesc, ydesc
    try:
        probe("mixer_y420p tick (1080p, 3 layers)", layers, n)
    finally:
        tdesc = keep


for rows, desc in ((None, None), (None, "host"), ("16", None)):
    cv.set_switch("CHV_WAVE_ROWS", rows)
    cv.set_switch("CHV_DESC", desc)
    print(f"-- CHV_WAVE_ROWS={rows} CHV_DESC={desc} (descriptors of a transient launch: copied to device memory | read from the pinned host ring)")
    probe("empty tick (clear only)", [])
    probe("cfg2 tick (1 NV12 layer)", four[:1])
    cv.set_switch("CHV_BGRA_PATH", "wave")
    probe("cfg2 tick through the wave kernel", four[:1])
    cv.set_switch("CHV_BGRA_PATH", None)
    probe("2 NV12 layers", four[:2])
    probe("pipeline tick (4 NV12 layers)", four)
    probe_yuv()
    cv.set_switch("CHV_SAME_GEOM", "0")
    probe("pipeline tick, no geometry sharing", four)
    cv.set_switch("CHV_SAME_GEOM", None)
