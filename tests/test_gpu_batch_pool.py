"""A batch's descriptors live in a block of a per-device pool (chipvideo.cpp: DescBlock — device memory, a pinned twin, an event): creation sends
them with ONE asynchronous copy on the creating context's stream, every run marks the block with its event, a block taken again waits for that
event before its twin is overwritten.  What a host that builds a batch per group tick relies on: destroying a batch whose launch is still
running and building the next one in the same block must not touch the first one's canvases; a batch created on one context runs on another
(its copy went out on the creator's stream); blocks of every size class come back."""
import numpy as np
import pytest

import gpuutil as G
import util
from oracle import oracle as O
from swiftvideo_amd import chipvideo as cv
from swiftvideo_amd import compute as sv

pytestmark = pytest.mark.gpu


def _ticks(ctx, n, seed, cw=320, ch=180, sw=480, sh=270, layers=2):
    """n ticks of `layers` NV12 videos on BGRA canvases: ([(target, clear, layers)], [oracle canvas], [target], keep)"""
    out, exps, gds, keep = [], [], [], []
    for t in range(n):
        exp = util.alloc_image("bgra", cw, ch, seed=seed + t)
        assert O.run_kernel("img_clear_bgra", exp) == 0
        ls = []
        for l in range(layers):
            src = util.alloc_image("nv12", sw, sh, seed=seed * 13 + 5 * t + l)
            u = util.full_canvas_uniforms((cw, ch), (sw, sh), opacity=1.0 - 0.3 * l)
            assert O.run_kernel("img_nv12_bgra", exp, src, u, threads=4) == 0
            g = G.to_gpu(ctx, "nv12", sw, sh, src)
            keep.append(g)
            ls.append((sv.ComputeKernel.img_nv12_bgra, g, u, 0))
        gd = G.to_gpu(ctx, "bgra", cw, ch, util.alloc_image("bgra", cw, ch, seed=seed + t))
        out.append((gd, True, ls)); exps.append(exp); gds.append(gd)
    return out, exps, gds, keep


def test_a_batch_destroyed_in_flight_and_the_next_one_in_its_block(ctx):
    """run without waiting, destroy at once, build and run the next group tick (same size: the same block comes back) — twenty times; every
    tick of every batch has the oracle's canvas at the end"""
    lib = cv.load()
    rounds = []
    for r in range(20):
        ticks, exps, gds, keep = _ticks(ctx, 6, 100 + 10 * r)
        h, name, ka = G.make_batch(ctx, ticks)
        cv.check(lib.chv_pass_begin(ctx.handle)); cv.check(lib.chv_batch_run(ctx.handle, h)); cv.check(lib.chv_pass_end(ctx.handle, 0))
        G.destroy_batch(h)
        rounds.append((exps, gds, keep, ka))
    cv.check(lib.chv_pass_begin(ctx.handle)); cv.check(lib.chv_pass_end(ctx.handle, 1))
    for r, (exps, gds, keep, ka) in enumerate(rounds):
        for i, (gd, exp) in enumerate(zip(gds, exps)):
            G.assert_same(G.from_gpu(ctx, gd, "bgra", 320, 180), exp, f"group tick {r}, tick {i}")


def test_a_batch_created_on_one_context_runs_on_another(ctx):
    """the descriptors' copy is on the creator's stream: a run on a sharing context's stream waits for the block's event first"""
    lib = cv.load()
    other = sv.createComputeContext(sharing=ctx)
    for r in range(5):
        ticks, exps, gds, keep = _ticks(ctx, 4, 400 + 10 * r)
        h, name, ka = G.make_batch(ctx, ticks)
        for c in (other, ctx, other):
            cv.check(lib.chv_pass_begin(c.handle)); cv.check(lib.chv_batch_run(c.handle, h)); cv.check(lib.chv_pass_end(c.handle, 1))
            for i, (gd, exp) in enumerate(zip(gds, exps)):
                G.assert_same(G.from_gpu(ctx, gd, "bgra", 320, 180), exp, f"round {r}, tick {i}")
        G.destroy_batch(h)


@pytest.mark.parametrize("n", [1, 3, 40, 200, 700])
def test_blocks_of_every_size_class(ctx, n):
    """1 tick (64 KB block) to 700 ticks of two layers (a 1 MB block), each size twice: from the system, then from the pool"""
    for rep in range(2):
        ticks, exps, gds, keep = _ticks(ctx, n, 900 + n + rep, cw=64, ch=32, sw=96, sh=48)
        h, name, ka = G.make_batch(ctx, ticks)
        G.run_batch(ctx, h)
        for i in sorted({0, n // 2, n - 1}):
            G.assert_same(G.from_gpu(ctx, gds[i], "bgra", 64, 32), exps[i], f"{n} ticks, rep {rep}, tick {i} ({name})")
        G.destroy_batch(h)
