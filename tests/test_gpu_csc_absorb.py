"""The absorbed form of the colour matrix on whole ticks (pixel_math.hip.h: csc_fold_absorbed; tests/test_csc_absorb.py for the table,
tests/test_gpu_matrices.py for every code triple): tick_bgra_stream picks its absorbed instantiation when EVERY layer's matrix has absorbing
biases and the plain one otherwise (BT.601 full range has none); the tiled kernel's fast rows need them and the host sends a BT.601 full-range
layer to the strip kernel — forced onto the tiled kernel it takes that kernel's per-pixel rows.  Every combination must be the oracle's bytes,
on NV12 and planar sources."""
import numpy as np
import pytest

import gpuutil as G
import util
from oracle import oracle as O
from swiftvideo_amd import compute as sv

pytestmark = pytest.mark.gpu

CW, CH, SW, SH = 320, 180, 480, 270


def _tick(ctx, fmt, cscs, opacities, seed):
    exp = util.alloc_image("bgra", CW, CH)
    assert O.run_kernel("img_clear_bgra", exp) == 0
    layers = []
    for i, (c, o) in enumerate(zip(cscs, opacities)):
        u = util.make_uniforms((CW, CH), in_size=(SW, SH), opacity=o)
        src = util.alloc_image(fmt, SW, SH, seed=seed + i)
        assert O.run_kernel(f"img_{fmt}_bgra", exp, src, u, csc=c, threads=4) == 0
        layers.append((sv.defaultComputeKernelFromString(f"img_{fmt}_bgra"), G.to_gpu(ctx, fmt, SW, SH, src), u, c))
    gd = G.to_gpu(ctx, "bgra", CW, CH, util.alloc_image("bgra", CW, CH, seed=seed + 50))
    h, name, keep = G.make_batch(ctx, [(gd, True, layers)])
    G.run_batch(ctx, h)
    G.destroy_batch(h)
    G.assert_same(G.from_gpu(ctx, gd, "bgra", CW, CH), exp, f"{fmt} csc {cscs} via {name}")
    return name


@pytest.mark.parametrize("fmt", ["nv12", "y420p"])
@pytest.mark.parametrize("cscs", [(0, 0, 0, 0), (1, 1), (3, 3, 3), (2, 2), (0, 1, 3, 1), (3, 0, 2, 1), (2, 0), (1, 3)], ids=lambda c: "csc" + "".join(map(str, c)))
def test_stream_ticks_in_every_colourspace(ctx, fmt, cscs):
    name = _tick(ctx, fmt, cscs, (1.0, 0.75, 0.5, 0.25), seed=700 + sum(cscs))
    assert name == "tick_bgra_stream", name


@pytest.mark.parametrize("fmt", ["nv12", "y420p"])
@pytest.mark.parametrize("csc", [0, 1, 2, 3])
def test_one_layer_ticks_in_every_colourspace(ctx, fmt, csc, switch):
    # the library's own route: the tiled kernel, or — BT.601 full range — the strip kernel
    name = _tick(ctx, fmt, (csc,), (1.0,), seed=720 + csc)
    assert name == ("tick_bgra_wave" if csc == 2 else f"tick_{fmt}_bgra_tiled"), name
    # forced onto the tiled kernel: its per-pixel rows for the matrix without absorbing biases
    switch("CHV_BGRA_PATH", "tiled")
    name = _tick(ctx, fmt, (csc,), (1.0,), seed=730 + csc)
    assert name == f"tick_{fmt}_bgra_tiled", name
    # ... and through the streaming kernel's one-layer instantiations
    switch("CHV_BGRA_PATH", "stream")
    name = _tick(ctx, fmt, (csc,), (0.6,), seed=740 + csc)
    assert name == "tick_bgra_stream", name


@pytest.mark.parametrize("cscs", [(0, 1, 3), (2, 1)], ids=lambda c: "csc" + "".join(map(str, c)))
def test_batches_of_many_ticks_choose_per_launch(ctx, cscs):
    """ONE tick with a BT.601 full-range layer puts the whole launch on the plain-matrix instantiation: still the oracle's bytes for every tick"""
    rng = np.random.default_rng(len(cscs))
    ticks, exps, gds = [], [], []
    for t in range(6):
        exp = util.alloc_image("bgra", CW, CH)
        assert O.run_kernel("img_clear_bgra", exp) == 0
        layers = []
        for i in range(2):
            c = int(cscs[(t + i) % len(cscs)])
            u = util.make_uniforms((CW, CH), in_size=(SW, SH), opacity=(1.0, 0.5)[i])
            src = util.alloc_image("nv12", SW, SH, seed=int(rng.integers(1, 1 << 20)))
            assert O.run_kernel("img_nv12_bgra", exp, src, u, csc=c, threads=4) == 0
            layers.append((sv.ComputeKernel.img_nv12_bgra, G.to_gpu(ctx, "nv12", SW, SH, src), u, c))
        gd = G.to_gpu(ctx, "bgra", CW, CH, util.alloc_image("bgra", CW, CH, seed=60 + t))
        ticks.append((gd, True, layers)); exps.append(exp); gds.append(gd)
    h, name, keep = G.make_batch(ctx, ticks)
    assert name == "tick_bgra_stream", name
    G.run_batch(ctx, h)
    G.destroy_batch(h)
    for i, (gd, exp) in enumerate(zip(gds, exps)):
        G.assert_same(G.from_gpu(ctx, gd, "bgra", CW, CH), exp, f"tick {i}")


@pytest.mark.parametrize("rows, streamed", [(32766, True), (32770, False)])
def test_stream_row_table_holds_the_tallest_planes_it_admits(ctx, rows, streamed):
    """The streaming kernel's row table packs a row's two tap rows into 16 / 15 signed bits (kernels_stream.hip.cpp): planes of up to 32767 luma /
    16383 chroma rows are admitted — the last rows of such a picture are the largest values the fields ever hold — taller ones go to the strip
    kernel.  Two same-geometry NV12 layers, 4 : 1 down onto an 8200-row canvas."""
    cw, ch, sw = 64, (rows + 3) // 4 + 8, 64
    exp = util.alloc_image("bgra", cw, ch)
    assert O.run_kernel("img_clear_bgra", exp) == 0
    layers = []
    for i, o in enumerate((1.0, 0.5)):
        u = util.make_uniforms((cw, ch), in_size=(sw, rows), opacity=o)
        src = util.alloc_image("nv12", sw, rows, seed=810 + i)
        assert O.run_kernel("img_nv12_bgra", exp, src, u, threads=8) == 0
        layers.append((sv.ComputeKernel.img_nv12_bgra, G.to_gpu(ctx, "nv12", sw, rows, src), u, 0))
    gd = G.to_gpu(ctx, "bgra", cw, ch, util.alloc_image("bgra", cw, ch, seed=820))
    h, name, keep = G.make_batch(ctx, [(gd, True, layers)])
    assert (name == "tick_bgra_stream") == streamed, name
    G.run_batch(ctx, h)
    G.destroy_batch(h)
    G.assert_same(G.from_gpu(ctx, gd, "bgra", cw, ch), exp, f"{rows} source rows via {name}")
