"""tools/latency_probe.py — per-tick cost of the single-tick path a mixer uses (chv_composite, one launch per tick):
asynchronous submission with one wait at the end, and the reference's own pattern of a wait after every tick
(usingContext, compute.swift:131-134).  cfg2 geometry (1080p NV12 -> 720p BGRA).  Run on the GPU box."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, ctypes as C
import util, gpuutil as G
from swiftvideo_amd import compute as sv, chipvideo as cv
ctx = sv.makeComputeContext(forType="GPU")
src = G.to_gpu(ctx, "nv12", 1920, 1080, util.alloc_image("nv12", 1920, 1080, seed=1))
dst = G.to_gpu(ctx, "bgra", 1280, 720, util.alloc_image("bgra", 1280, 720))
u = util.full_canvas_uniforms((1280, 720), (1920, 1080))
full = src.derive(matrix=sv._unit_quad_to_ndc(), borderMatrix=sv._unit_quad_to_ndc())
lib = cv.load()
tdesc = sv._image_desc(dst)
def probe(label, layers):
    arr = sv._layer_array(layers)
    def run(n):
        for _ in range(n):
            lib.chv_composite(ctx.handle, C.byref(tdesc), 1, arr, len(layers))
        lib.chv_pass_end(ctx.handle, 1)
    run(200)
    t = time.perf_counter(); run(2000); dt = time.perf_counter() - t
    print(f"{label}: chv_composite x2000: {dt / 2000 * 1e6:.1f} us per tick (async submit, one sync at the end)")
    t = time.perf_counter()
    for _ in range(500):
        lib.chv_composite(ctx.handle, C.byref(tdesc), 1, arr, len(layers)); lib.chv_pass_end(ctx.handle, 1)
    print(f"{label}: chv_composite + wait: {(time.perf_counter() - t) / 500 * 1e6:.1f} us per tick")
probe("cfg2 tick (1 NV12 layer)", [(sv.ComputeKernel.img_nv12_bgra, full, sv.imageUniformsFor(full, dst), 0)])
# the headline tick as a mixer issues it: four 1080p NV12 streams, opacity 1 / .75 / .5 / .25, one launch
srcs = [G.to_gpu(ctx, "nv12", 1920, 1080, util.alloc_image("nv12", 1920, 1080, seed=2 + i)) for i in range(4)]
four = []
for s4, o in zip(srcs, (1.0, 0.75, 0.5, 0.25)):
    f = s4.derive(matrix=sv._unit_quad_to_ndc(), borderMatrix=sv._unit_quad_to_ndc(), opacity=o)
    four.append((sv.ComputeKernel.img_nv12_bgra, f, sv.imageUniformsFor(f, dst), 0))
probe("pipeline tick (4 NV12 layers)", four)
