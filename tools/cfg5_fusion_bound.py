"""tools/cfg5_fusion_bound.py — what a fused composite -> Lanczos band kernel could gain on cfg5 (8 x 2160p BGRA layers -> 2160p canvas ->
Lanczos-3 -> 1080p), MEASURED from the two kernels that exist instead of argued (run on the GPU box; prints a table).

A fused kernel composites a band of 2R + 10 canvas rows per R output rows into LDS and filters from there: the composite's arithmetic grows
by the band overlap (R = 32: 74 / 64 = 1.16), the canvas never travels (no 2160p stores, no Lanczos loads from HBM), one launch less per
frame.  Its best case is therefore
    t_fused >= overlap x t_composite(without its canvas stores' cost) + t_lanczos(source already on chip)
which this script brackets with things that can be timed today:
    a  composite alone (24 frames, distinct canvases)                         -> t_composite
    b  Lanczos alone, distinct canvases (HBM-resident source)                 -> t_lanczos_hbm
    c  Lanczos alone, every pair reading ONE canvas (cache-resident source)   -> t_lanczos_chip   (what "the canvas never travels" is worth to it)
    d  composite with every tick writing ONE canvas (stores hit the cache)    -> t_composite_chip (what "no canvas stores" is worth to it)
"""
import ctypes as C
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import bench  # noqa: E402
from swiftvideo_amd import chipvideo as cv  # noqa: E402
from swiftvideo_amd import compute as sv  # noqa: E402

lib = cv.load()
ctx = sv.makeComputeContext(forType="GPU")
dev = bench.HipDevice(cv, lib, ctx)
wl = bench.WORKLOADS["cfg5"]


def time_ms(fn, seconds=0.5):
    e0, e1 = dev.event(), dev.event()
    fn(); dev.sync()
    reps = 2
    while True:
        dev.record(e0)
        for _ in range(reps):
            fn()
        dev.record(e1); dev.sync()
        el = dev.elapsed_ms(e0, e1)
        if el >= seconds * 1e3:
            break
        reps = max(reps * 2, int(reps * seconds * 1e3 / max(el, 1e-3)) + 1)
    dev.destroy(e0); dev.destroy(e1)
    return el / reps


out = {}
w = bench.build_workload(sv, ctx, wl, wl["frames"], seed_base=0x5EED0000 + 80)
lz = sv.LanczosBatch(w["lanczos"])
out["a_composite"] = time_ms(lambda: cv.check(lib.chv_batch_run(ctx.handle, w["batch"])))
out["b_lanczos_hbm"] = time_ms(lambda: lz.run(ctx))
out["both"] = time_ms(lambda: (cv.check(lib.chv_batch_run(ctx.handle, w["batch"])), lz.run(ctx)))
lz1 = sv.LanczosBatch([(dst, w["lanczos"][0][1]) for dst, _ in w["lanczos"]])
out["c_lanczos_source_on_chip"] = time_ms(lambda: lz1.run(ctx))
bench.free_workload(w)
w = bench.build_workload(sv, ctx, wl, wl["frames"], seed_base=0x5EED0000 + 80, alias="dst")
out["d_composite_stores_on_chip"] = time_ms(lambda: cv.check(lib.chv_batch_run(ctx.handle, w["batch"])))
bench.free_workload(w)
overlap = 74.0 / 64.0
out["fused_best_case"] = overlap * out["d_composite_stores_on_chip"] + out["c_lanczos_source_on_chip"]
out["fused_best_case_vs_two_launches"] = out["fused_best_case"] / out["both"] - 1.0
print(json.dumps({k: round(v, 4) for k, v in out.items()}))
