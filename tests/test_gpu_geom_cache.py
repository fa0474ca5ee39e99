"""The strip kernels' per-layer geometry in BATCHES comes from tables computed once per batch and launch configuration (csrc/geom_cache.h,
WaveStrip::setup_cached; the CACHED instantiations of tick_bgra_wave / tick_yuv_wave contain no set-up code).  The tables are filled by the
set-up code itself, so the bytes must be those of the kernels that compute their geometry in place (CHV_GEOM_CACHE=0: the A/B and the path of
every transient launch) and of the oracle — also when the launch configuration of a batch changes between two runs, when ticks of a batch
differ in canvas size, when a layer's rectangles do not fit the LDS (the unstaged form), and when the batch has hundreds of geometries."""
import numpy as np
import pytest

import gpuutil as G
import util
from oracle import oracle as O
from swiftvideo_amd import compute as sv
import test_gpu_mixpath as MIX
import test_gpu_yuvwave as YWV

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows", ["8", "16"])
@pytest.mark.parametrize("seed", range(300, 316))
def test_fuzz_strip_routes_with_geometry_computed_in_place(ctx, switch, rows, seed):
    """the strip kernels' random ticks (test_gpu_fuzz.py runs them with the tables) with CHV_GEOM_CACHE=0: the instantiations every transient launch takes"""
    switch("CHV_GEOM_CACHE", "0")
    switch("CHV_BGRA_PATH", "wave")
    switch("CHV_YUV_STREAM", "0")
    switch("CHV_WAVE_ROWS", rows)
    MIX.test_random_mixed_ticks(ctx, MIX.WAVE, seed)
    YWV.test_random_yuv_ticks(ctx, rows, seed)


def _grid_ticks(ctx, dst, n_ticks, sizes):
    """ticks of a 2 x 2 grid over a background (five geometries), canvases of `sizes` in turn"""
    src_fmt = "nv12" if dst == "bgra" else dst
    ticks, exps, gds, keep = [], [], [], []
    for t in range(n_ticks):
        cw, ch = sizes[t % len(sizes)]
        sw, sh = cw * 3 // 2 // 2 * 2, ch * 3 // 2 // 2 * 2
        qw, qh = cw // 2, ch // 2
        us = [util.full_canvas_uniforms((cw, ch), (sw, sh))] + \
             [util.make_uniforms((cw, ch), rect=(qx * qw, qy * qh, qw, qh), opacity=o, in_size=(sw, sh)) for (qx, qy), o in zip(((0, 0), (1, 0), (0, 1), (1, 1)), (1.0, 0.9, 0.8, 0.7))]
        exp = util.alloc_image(dst, cw, ch, seed=40 + t)
        assert O.run_kernel(f"img_clear_{dst}", exp) == 0
        layers = []
        for l, u in enumerate(us):
            src = util.alloc_image(src_fmt, sw, sh, seed=1000 + 7 * t + l)
            name = f"img_{src_fmt}_{dst}"
            assert O.run_kernel(name, exp, src, u, threads=4) == 0
            g = G.to_gpu(ctx, src_fmt, sw, sh, src)
            keep.append(g)
            layers.append((sv.defaultComputeKernelFromString(name), g, u, 0))
        gd = G.to_gpu(ctx, dst, cw, ch, util.alloc_image(dst, cw, ch, seed=40 + t))
        ticks.append((gd, True, layers)); exps.append(exp); gds.append((gd, cw, ch))
    return ticks, exps, gds, keep


@pytest.mark.parametrize("dst", ["bgra", "nv12", "y420p"])
def test_a_batch_rebuilds_its_tables_when_the_launch_configuration_changes(ctx, switch, dst):
    """one batch, run with 8-row strips, with 16-row strips, with the tables off, with them on again: every run gives the oracle's canvases
    (the tables are keyed by strip height and LDS layout; the device layers lose and regain their table pointers)"""
    ticks, exps, gds, keep = _grid_ticks(ctx, dst, 6, [(256, 144), (192, 96)])
    if dst == "bgra":
        switch("CHV_BGRA_PATH", "wave")
    switch("CHV_YUV_STREAM", "0")
    h, name, ka = G.make_batch(ctx, ticks)
    assert "wave" in name, name
    for step, (sw_name, sw_val) in enumerate([("CHV_WAVE_ROWS", "8"), ("CHV_WAVE_ROWS", "16"), ("CHV_GEOM_CACHE", "0"), ("CHV_GEOM_CACHE", None), ("CHV_WAVE_ROWS", "8"), ("CHV_GEOM_CACHE", "eager"),
                                                ("CHV_WAVE_ROWS", "16")]):
        switch(sw_name, sw_val)
        for rerun in range(3):          # (the first run with a configuration computes in place, the tables are built at the second)
            G.run_batch(ctx, h)
            for i, ((gd, cw, ch), exp) in enumerate(zip(gds, exps)):
                G.assert_same(G.from_gpu(ctx, gd, dst, cw, ch), exp, f"{dst} step {step} ({sw_name}={sw_val}) run {rerun} tick {i}")
    G.destroy_batch(h)


def test_rectangles_that_do_not_fit_the_lds_take_the_unstaged_form_from_the_table(ctx, switch):
    """a 40:1 horizontal reduction next to ordinary layers: that layer's rectangles are far wider than a wave's LDS region, its strips are flagged
    unstaged in the table (tap positions instead of LDS offsets) and sampled from global memory — inside the cached kernel, in z order"""
    cw, ch = 128, 48
    wide = util.alloc_image("bgra", 5120, 32, seed=77)
    vid = util.alloc_image("nv12", 192, 72, seed=78)
    uw = util.make_uniforms((cw, ch), rect=(0, 8, 128, 32), opacity=0.8, in_size=(5120, 32))
    uv = util.full_canvas_uniforms((cw, ch), (192, 72))
    results = []
    for mode in ("1", "0"):
        switch("CHV_GEOM_CACHE", mode)
        exps, ticks, gds = [], [], []
        gw, gv = G.to_gpu(ctx, "bgra", 5120, 32, wide), G.to_gpu(ctx, "nv12", 192, 72, vid)
        for t in range(3):
            exp = util.alloc_image("bgra", cw, ch, seed=5)
            assert O.run_kernel("img_clear_bgra", exp) == 0 and O.run_kernel("img_nv12_bgra", exp, vid, uv) == 0 and O.run_kernel("img_bgra_bgra_tx", exp, wide, uw) == 0
            gd = G.to_gpu(ctx, "bgra", cw, ch, util.alloc_image("bgra", cw, ch, seed=5))
            ticks.append((gd, True, [(sv.ComputeKernel.img_nv12_bgra, gv, uv, 0), (sv.ComputeKernel.img_bgra_bgra_tx, gw, uw, 0)]))
            exps.append(exp); gds.append(gd)
        h, name, ka = G.make_batch(ctx, ticks)
        for run in range(3):
            G.run_batch(ctx, h)
            for i, (gd, exp) in enumerate(zip(gds, exps)):
                G.assert_same(G.from_gpu(ctx, gd, "bgra", cw, ch), exp, f"tables {mode}, run {run}, tick {i}, {name}")
        G.destroy_batch(h)
        results.append(name)
    assert results[0] == results[1]


def test_a_batch_of_hundreds_of_geometries_keeps_computing_them_in_place(ctx, switch):
    """more than 256 distinct geometries in a batch: tables that are each used once are not built (geom_cache_prepare), the kernels compute"""
    switch("CHV_BGRA_PATH", "wave")
    rng = np.random.default_rng(4)
    src = util.alloc_image("nv12", 96, 54, seed=9)
    g = G.to_gpu(ctx, "nv12", 96, 54, src)
    cw, ch = 128, 64
    ticks, exps, gds = [], [], []
    for t in range(300):
        u = util.make_uniforms((cw, ch), rect=(float(rng.uniform(0, 60)), float(rng.uniform(0, 30)), float(rng.uniform(30, 90)), float(rng.uniform(20, 40))), in_size=(96, 54))
        exp = util.alloc_image("bgra", cw, ch, seed=3)
        assert O.run_kernel("img_clear_bgra", exp) == 0 and O.run_kernel("img_nv12_bgra", exp, src, u) == 0
        gd = G.to_gpu(ctx, "bgra", cw, ch, util.alloc_image("bgra", cw, ch, seed=3))
        ticks.append((gd, True, [(sv.ComputeKernel.img_nv12_bgra, g, u, 0)])); exps.append(exp); gds.append(gd)
    h, name, ka = G.make_batch(ctx, ticks)
    for _ in range(3):
        G.run_batch(ctx, h)
    G.destroy_batch(h)
    for i in (0, 1, 150, 299):
        G.assert_same(G.from_gpu(ctx, gds[i], "bgra", cw, ch), exps[i], f"tick {i} of 300 through {name}")
