"""Launches of the strip kernels that CONTINUE on composed canvases (the second launch of a split batch — a logo or overlays over the videos the
streaming kernel did; layers added to a canvas outside a clear) cover only the strips the union of their layers' bounding boxes touches
(launch_wave_layers, kernels_wave_yuv.hip.cpp: the origin of the grid travels in the high halves of the two row counts).  Everything outside that
box must keep its bytes, everything inside must be the oracle's — for boxes in every corner, across strip boundaries, hanging over the canvas
edge, rotated, and for batches whose ticks have their layers in different places."""
import numpy as np
import pytest

import gpuutil as G
import util
from oracle import oracle as O
from swiftvideo_amd import compute as sv

pytestmark = pytest.mark.gpu

CW, CH = 448, 200           # 7 x 25 strips of 64 x 8


def _layers_for(dst, rng, n):
    srcs = {"bgra": ["bgra", "rgba", "nv12"], "nv12": ["bgra", "nv12"], "y420p": ["bgra", "y420p"]}[dst]
    out = []
    for _ in range(n):
        s = srcs[int(rng.integers(0, len(srcs)))]
        name = f"img_{s}_{dst}" + ("_tx" if dst == "bgra" and s in ("bgra", "rgba") else "")
        iw, ih = int(rng.integers(8, 60)) * 2, int(rng.integers(8, 40)) * 2
        out.append((name, s, iw, ih))
    return out


RECTS = [  # (x, y, w, h, rotation): corners, strip-aligned, straddling, over the edges, a point-sized box, the whole canvas
    (0, 0, 40, 20, 0.0), (CW - 50, CH - 30, 50, 30, 0.0), (64, 8, 64, 8, 0.0), (63, 7, 66, 10, 0.0), (200, 90, 30, 30, 0.7),
    (-30, -20, 80, 60, 0.0), (CW - 20, 100, 90, 40, 0.0), (300, -10, 100, 30, 0.2), (129, 65, 2, 2, 0.0), (0, 0, CW, CH, 0.0),
    (20, CH - 9, 400, 18, 0.0), (CW - 1, CH - 1, 30, 30, 0.0),
]


@pytest.mark.parametrize("dst", ["bgra", "nv12", "y420p"])
@pytest.mark.parametrize("rect", range(len(RECTS)))
def test_uncleared_launch_touches_only_its_layers_box(ctx, dst, rect):
    x, y, w, h, rot = RECTS[rect]
    rng = np.random.default_rng(100 * rect + len(dst))
    (name, s, iw, ih), = _layers_for(dst, rng, 1)
    u = util.make_uniforms((CW, CH), rect=(x, y, w, h), rotation=rot, opacity=0.8, in_size=(iw, ih))
    src = util.alloc_image(s, iw, ih, seed=7 + rect)
    canvas0 = util.alloc_image(dst, CW, CH, seed=11 + rect)
    exp = util.copy_image(canvas0)
    assert O.run_kernel(name, exp, src, u) == 0
    gd, gs = G.to_gpu(ctx, dst, CW, CH, canvas0), G.to_gpu(ctx, s, iw, ih, src)      # (a batch borrows its pictures: both stay alive until it has run)
    h_, kname, keep = G.make_batch(ctx, [(gd, False, [(sv.defaultComputeKernelFromString(name), gs, u, 0)])])
    G.run_batch(ctx, h_)
    G.destroy_batch(h_)
    G.assert_same(G.from_gpu(ctx, gd, dst, CW, CH), exp, f"{name} at {RECTS[rect]} through {kname}")
    del gs


@pytest.mark.parametrize("dst", ["bgra", "y420p"])
@pytest.mark.parametrize("seed", range(6))
def test_batch_whose_ticks_have_their_layers_in_different_places(ctx, dst, seed):
    """the grid is the UNION over the batch's ticks; ticks of different canvas sizes included"""
    rng = np.random.default_rng(300 + seed)
    ticks, exps, gds = [], [], []
    for t in range(6):
        cw, ch = (CW, CH) if t % 2 == 0 else (320, 96)
        canvas0 = util.alloc_image(dst, cw, ch, seed=50 + t)
        exp = util.copy_image(canvas0)
        layers = []
        for name, s, iw, ih in _layers_for(dst, rng, int(rng.integers(0, 3))):
            u = util.make_uniforms((cw, ch), rect=(float(rng.uniform(-40, cw)), float(rng.uniform(-20, ch)), float(rng.uniform(4, 120)), float(rng.uniform(4, 60))),
                                   rotation=float(rng.choice([0.0, 0.0, rng.uniform(-1, 1)])), opacity=float(rng.uniform(0.2, 1.0)), in_size=(iw, ih))
            src = util.alloc_image(s, iw, ih, seed=int(rng.integers(1, 1 << 20)))
            assert O.run_kernel(name, exp, src, u) == 0
            layers.append((sv.defaultComputeKernelFromString(name), G.to_gpu(ctx, s, iw, ih, src), u, 0))
        gd = G.to_gpu(ctx, dst, cw, ch, canvas0)
        ticks.append((gd, False, layers)); exps.append(exp); gds.append((gd, cw, ch))
    if all(not t[2] for t in ticks):
        pytest.skip("no layer in any tick of this seed")
    h_, kname, keep = G.make_batch(ctx, ticks)
    G.run_batch(ctx, h_)
    G.destroy_batch(h_)
    for i, ((gd, cw, ch), exp) in enumerate(zip(gds, exps)):
        G.assert_same(G.from_gpu(ctx, gd, dst, cw, ch), exp, f"tick {i} of a {kname} batch, seed {seed}")


def test_split_batch_second_launch_covers_the_logo_only(ctx):
    """the shape the bounding-box grid is for: 'two same-geometry videos, then a small rotated logo' = tick_bgra_stream + tick_bgra_wave"""
    cw, ch, sw, sh = 640, 360, 960, 540
    vids = [util.alloc_image("nv12", sw, sh, seed=900 + i) for i in range(2)]
    logo = util.alloc_image("rgba", 80, 44, seed=910)
    us = [util.full_canvas_uniforms((cw, ch), (sw, sh)), util.full_canvas_uniforms((cw, ch), (sw, sh), opacity=0.5)]
    ul = util.make_uniforms((cw, ch), rect=(500, 30, 80, 44), rotation=0.3, opacity=0.9, in_size=(80, 44))
    gv = [G.to_gpu(ctx, "nv12", sw, sh, v) for v in vids]
    gl = G.to_gpu(ctx, "rgba", 80, 44, logo)
    ticks, exps, gds = [], [], []
    for t in range(9):
        exp = util.alloc_image("bgra", cw, ch)
        assert O.run_kernel("img_clear_bgra", exp) == 0
        for v, u in zip(vids, us):
            assert O.run_kernel("img_nv12_bgra", exp, v, u, threads=4) == 0
        assert O.run_kernel("img_rgba_bgra_tx", exp, logo, ul) == 0
        gd = G.to_gpu(ctx, "bgra", cw, ch, util.alloc_image("bgra", cw, ch, seed=920 + t))
        ticks.append((gd, True, [(sv.ComputeKernel.img_nv12_bgra, gv[0], us[0], 0), (sv.ComputeKernel.img_nv12_bgra, gv[1], us[1], 0),
                                 (sv.ComputeKernel.img_rgba_bgra_tx, gl, ul, 0)]))
        exps.append(exp); gds.append(gd)
    h_, kname, keep = G.make_batch(ctx, ticks)
    assert kname == "tick_bgra_stream + tick_bgra_wave", kname
    G.run_batch(ctx, h_)
    G.destroy_batch(h_)
    for i, (gd, exp) in enumerate(zip(gds, exps)):
        G.assert_same(G.from_gpu(ctx, gd, "bgra", cw, ch), exp, f"tick {i}")


@pytest.mark.parametrize("rect", range(len(RECTS)))
def test_general_kernel_on_an_uncleared_canvas_covers_its_layers_boxes(ctx, rect, switch):
    """tick_general_bgra too (launch_tick_general: the grid of a launch that clears nothing is the union of its layers' bounding boxes)"""
    switch("CHV_FORCE_GENERAL", "1")
    x, y, w, h, rot = RECTS[rect]
    rng = np.random.default_rng(500 + rect)
    (name, s, iw, ih), = _layers_for("bgra", rng, 1)
    u = util.make_uniforms((CW, CH), rect=(x, y, w, h), rotation=rot, opacity=0.7, in_size=(iw, ih))
    u2 = util.make_uniforms((CW, CH), rect=(CW - x - w, CH - y - h, w, h), rotation=-rot, opacity=0.5, in_size=(iw, ih))
    src = util.alloc_image(s, iw, ih, seed=17 + rect)
    canvas0 = util.alloc_image("bgra", CW, CH, seed=19 + rect)
    exp = util.copy_image(canvas0)
    assert O.run_kernel(name, exp, src, u) == 0
    assert O.run_kernel(name, exp, src, u2) == 0
    gd, gs = G.to_gpu(ctx, "bgra", CW, CH, canvas0), G.to_gpu(ctx, s, iw, ih, src)
    k = sv.defaultComputeKernelFromString(name)
    h_, kname, keep = G.make_batch(ctx, [(gd, False, [(k, gs, u, 0), (k, gs, u2, 0)])])
    assert kname == "tick_general_bgra", kname
    G.run_batch(ctx, h_)
    G.destroy_batch(h_)
    G.assert_same(G.from_gpu(ctx, gd, "bgra", CW, CH), exp, f"{name} at {RECTS[rect]} and mirrored through {kname}")
    del gs
