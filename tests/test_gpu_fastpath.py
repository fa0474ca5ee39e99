"""The axis-aligned LDS-tiled kernels give exactly the bytes of the oracle (and therefore of
the general kernels), across tile edges, scale factors, partial cover, borders, fill, opacity,
flips, odd sizes and the LDS-overflow fallback; and the host really selects them."""
import numpy as np
import pytest

import gpuutil as G
import util
from oracle import oracle as O
from swiftvideo_amd import compute as sv

pytestmark = pytest.mark.gpu

# (canvas w, h, src w, h, make_uniforms kwargs, clear_first)
NV12_BGRA_CASES = {
    "cfg2_small":     (320, 180, 480, 270, dict(), True),
    "same_size":      (200, 60, 200, 60, dict(), True),
    "upscale_3x":     (300, 90, 100, 30, dict(opacity=0.5), True),
    "down_2.5x":      (130, 50, 326, 124, dict(), True),
    "rect_border":    (260, 70, 96, 54, dict(rect=(33, 9, 180, 40), border=(5, 3, 7, 2), fill=(0.9, 0.2, 0.1, 0.6), opacity=0.8), True),
    "noclear_opaque": (260, 70, 96, 54, dict(rect=(33, 9, 180, 40)), False),
    "noclear_blend":  (260, 70, 96, 54, dict(rect=(-20, -10, 200, 100), opacity=0.35, fill=(0.1, 0.5, 0.9, 1.0), border=(40, 40, 40, 40)), False),
    "tex_window":     (257, 33, 120, 80, dict(tex=(0.2, 0.1, 0.5, 0.7)), True),
    "letterbox":      (256, 48, 64, 64, dict(rect=(40, 4, 180, 40), tex=(0.25, 0.0, 0.5, 1.0), fill=(1, 1, 0, 1)), True),
    "flip_x":         (192, 40, 96, 54, dict(tex=(1.0, 0.0, -1.0, 1.0)), True),
    "odd_width":      (131, 19, 97, 41, dict(), True),
    "tiny":           (3, 2, 8, 6, dict(), True),
    "huge_downscale": (64, 20, 2048, 640, dict(), True),   # tile does not fit LDS -> unstaged taps or general path
}


@pytest.mark.parametrize("case", list(NV12_BGRA_CASES))
@pytest.mark.parametrize("csc", [0, 3])
def test_nv12_bgra_tiled_matches_oracle(ctx, case, csc):
    cw, ch, sw, sh, kw, clear = NV12_BGRA_CASES[case]
    u = util.make_uniforms((cw, ch), in_size=(sw, sh), **kw)
    src = util.alloc_image("nv12", sw, sh, seed=21)
    canvas0 = util.alloc_image("bgra", cw, ch, seed=22)
    exp = util.copy_image(canvas0)
    if clear:
        assert O.run_kernel("img_clear_bgra", exp) == 0
    assert O.run_kernel("img_nv12_bgra", exp, src, u, csc=csc, threads=4) == 0
    gs = G.to_gpu(ctx, "nv12", sw, sh, src)
    gd = G.to_gpu(ctx, "bgra", cw, ch, canvas0)
    h, name, keep = G.make_batch(ctx, [(gd, clear, [(sv.ComputeKernel.img_nv12_bgra, gs, u, csc)])])
    if case not in ("huge_downscale", "down_2.5x", "tiny"):   # those exceed the LDS tile budget -> general kernel
        assert name == "tick_nv12_bgra_tiled", f"axis-aligned NV12->BGRA batch dispatched to {name}"
    G.run_batch(ctx, h)
    G.destroy_batch(h)
    G.assert_same(G.from_gpu(ctx, gd, "bgra", cw, ch), exp, f"{case}/csc{csc} via {name}")


# 32-row tiles (chosen for launches of >= 1024 blocks; CHV_TILE_ROWS=32 forces them here): same bytes, including
# rectangles at a picture edge that overflow the prefetch registers and finish through stage_tail
TILE32_CASES = {
    "cfg2_small":   (320, 180, 480, 270, dict()),
    "tail":         (256, 96, 432, 164, dict()),          # 1.69 x 1.71: 58 rows x 14-16 vectors > 768 luma slots: the rest goes through stage_tail
    "partial_rows": (300, 75, 300, 75, dict(opacity=0.6)),
    "rect_border":  (260, 100, 96, 54, dict(rect=(33, 9, 180, 70), border=(5, 3, 7, 2), fill=(0.9, 0.2, 0.1, 0.6), opacity=0.8)),
    "upscale":      (384, 128, 128, 48, dict()),
}


@pytest.mark.parametrize("case", list(TILE32_CASES))
@pytest.mark.parametrize("fmt", ["nv12", "y420p"])
def test_yuv_bgra_32_row_tiles_match_oracle(ctx, switch, case, fmt):
    switch("CHV_TILE_ROWS", "32")
    cw, ch, sw, sh, kw = TILE32_CASES[case]
    u = util.make_uniforms((cw, ch), in_size=(sw, sh), **kw)
    src = util.alloc_image(fmt, sw, sh, seed=31)
    exp = util.alloc_image("bgra", cw, ch)
    kname = f"img_{fmt}_bgra"
    assert O.run_kernel("img_clear_bgra", exp) == 0
    assert O.run_kernel(kname, exp, src, u, threads=4) == 0
    gd = G.to_gpu(ctx, "bgra", cw, ch, util.alloc_image("bgra", cw, ch, seed=32))
    gs = G.to_gpu(ctx, fmt, sw, sh, src)          # (a batch borrows its pictures: kept alive until it has run)
    h, name, keep = G.make_batch(ctx, [(gd, True, [(sv.defaultComputeKernelFromString(kname), gs, u, 0)])])
    assert name == f"tick_{fmt}_bgra_tiled"
    G.run_batch(ctx, h)
    G.destroy_batch(h)
    G.assert_same(G.from_gpu(ctx, gd, "bgra", cw, ch), exp, f"{case}/{fmt} with 32-row tiles")


@pytest.mark.parametrize("seed", range(24))
@pytest.mark.parametrize("rows", ["16", "32"])
def test_random_axis_aligned_yuv_bgra_ticks(ctx, switch, seed, rows):
    """Seeded random axis-aligned geometry (placement, crop, flips, borders, fill, opacity, up- and downscales), three ticks of
    different sizes per launch, both tile heights: tiled kernels == oracle."""
    switch("CHV_TILE_ROWS", rows)
    rng = np.random.default_rng(7000 + seed)
    fmt = "nv12" if seed % 2 == 0 else "y420p"
    kname = f"img_{fmt}_bgra"
    clear = bool(rng.integers(0, 2))
    ticks, exps, gds = [], [], []
    for t in range(3):
        cw, ch = int(rng.integers(3, 330)), int(rng.integers(2, 140))
        sw, sh = int(rng.integers(8, 260)) * 2, int(rng.integers(2, 110)) * 2     # rows of >= 16 bytes for the staged loads
        kw = {}
        if rng.random() < 0.7:
            kw["rect"] = (float(rng.uniform(-0.3, 0.6) * cw), float(rng.uniform(-0.3, 0.6) * ch),
                          float(rng.uniform(0.2, 1.5) * cw), float(rng.uniform(0.2, 1.5) * ch))
        if rng.random() < 0.4:
            kw["border"] = tuple(float(v) for v in rng.uniform(0, 10, 4))
        if rng.random() < 0.4:
            kw["fill"] = tuple(float(v) for v in rng.uniform(0, 1, 4))
        if rng.random() < 0.5:
            kw["tex"] = (float(rng.uniform(0.0, 0.4)), float(rng.uniform(0.0, 0.4)),
                         float(rng.uniform(0.3, 1.0)) * (1 if rng.random() < 0.8 else -1), float(rng.uniform(0.3, 1.0)))
        kw["opacity"] = float(rng.choice([1.0, 1.0, rng.uniform(0, 1)]))
        u = util.make_uniforms((cw, ch), in_size=(sw, sh), **kw)
        src = util.alloc_image(fmt, sw, sh, seed=int(rng.integers(1, 1 << 20)))
        canvas0 = util.alloc_image("bgra", cw, ch, seed=int(rng.integers(1, 1 << 20)))
        exp = util.copy_image(canvas0)
        if clear:
            assert O.run_kernel("img_clear_bgra", exp) == 0
        assert O.run_kernel(kname, exp, src, u, csc=seed % 4, threads=4) == 0
        gd = G.to_gpu(ctx, "bgra", cw, ch, canvas0)
        ticks.append((gd, clear, [(sv.defaultComputeKernelFromString(kname), G.to_gpu(ctx, fmt, sw, sh, src), u, seed % 4)]))
        exps.append(exp)
        gds.append((gd, cw, ch))
    h, name, keep = G.make_batch(ctx, ticks)
    G.run_batch(ctx, h)
    G.destroy_batch(h)
    for i, ((gd, cw, ch), exp) in enumerate(zip(gds, exps)):
        G.assert_same(G.from_gpu(ctx, gd, "bgra", cw, ch), exp, f"seed {seed} rows {rows} tick {i} via {name}")


def test_rotated_layer_falls_back_to_general(ctx):
    u = util.make_uniforms((64, 36), rect=(10, 5, 40, 20), rotation=0.2, in_size=(32, 18))
    gs = G.to_gpu(ctx, "nv12", 32, 18, util.alloc_image("nv12", 32, 18, seed=1))
    gd = G.to_gpu(ctx, "bgra", 64, 36, util.alloc_image("bgra", 64, 36, seed=2))
    h, name, keep = G.make_batch(ctx, [(gd, True, [(sv.ComputeKernel.img_nv12_bgra, gs, u, 0)])])
    G.destroy_batch(h)
    assert name == "tick_general_bgra"


def test_batch_of_mixed_sizes(ctx):
    """One launch, ticks with different canvas/source sizes and geometry (grid.z-less XCD numbering)."""
    specs = [(320, 180, 480, 270, dict()), (130, 50, 326, 124, dict(opacity=0.7)), (64, 36, 64, 36, dict()),
             (257, 33, 120, 80, dict(tex=(0.2, 0.1, 0.5, 0.7))), (200, 60, 100, 30, dict(rect=(10, 10, 150, 40), border=(3, 3, 3, 3), fill=(0, 1, 0, 1))),
             (320, 180, 480, 270, dict()), (16, 16, 16, 16, dict()), (300, 90, 100, 30, dict()), (131, 19, 97, 41, dict()),
             (320, 180, 160, 90, dict()), (48, 200, 96, 400, dict())]
    ticks, exps, gds = [], [], []
    for i, (cw, ch, sw, sh, kw) in enumerate(specs):
        u = util.make_uniforms((cw, ch), in_size=(sw, sh), **kw)
        src = util.alloc_image("nv12", sw, sh, seed=50 + i)
        exp = util.alloc_image("bgra", cw, ch)
        assert O.run_kernel("img_clear_bgra", exp) == 0
        assert O.run_kernel("img_nv12_bgra", exp, src, u, threads=4) == 0
        gd = G.to_gpu(ctx, "bgra", cw, ch, util.alloc_image("bgra", cw, ch, seed=99))
        ticks.append((gd, True, [(sv.ComputeKernel.img_nv12_bgra, G.to_gpu(ctx, "nv12", sw, sh, src), u, 0)]))
        exps.append(exp)
        gds.append((gd, cw, ch))
    h, name, keep = G.make_batch(ctx, ticks)
    G.run_batch(ctx, h)
    G.run_batch(ctx, h)   # a batch can be replayed; clear_first makes it idempotent
    G.destroy_batch(h)
    for i, ((gd, cw, ch), exp) in enumerate(zip(gds, exps)):
        G.assert_same(G.from_gpu(ctx, gd, "bgra", cw, ch), exp, f"tick {i}")


# ---- BGRA/RGBA layers onto a BGRA canvas (tick_bgra_wave) ---------------------------------------------------
RGB_CASES = {
    # name: (canvas w, h, clear_first, [(kernel, src w, h, make_uniforms kwargs)])
    "one_full":      (200, 60, True, [("img_bgra_bgra_tx", 200, 60, dict())]),
    "cfg3_small":    (192, 108, True, [("img_bgra_bgra_tx", 192, 108, dict(opacity=o)) for o in (1.0, 0.75, 0.5, 0.25)]),
    "mixed_scale":   (260, 70, True, [("img_bgra_bgra_tx", 96, 54, dict()), ("img_rgba_bgra_tx", 400, 120, dict(opacity=0.6)),
                                        ("img_bgra_bgra_tx", 33, 17, dict(rect=(20, 10, 100, 40), opacity=0.9))]),
    "borders_fill":  (260, 70, True, [("img_bgra_bgra_tx", 64, 36, dict(rect=(33, 9, 180, 40), border=(5, 3, 7, 2), fill=(0.9, 0.2, 0.1, 0.6), opacity=0.8)),
                                        ("img_rgba_bgra_tx", 64, 36, dict(rect=(-20, -10, 120, 60), fill=(0.1, 0.5, 0.9, 1.0), border=(40, 40, 40, 40), opacity=0.35))]),
    "noclear":       (131, 39, False, [("img_bgra_bgra_tx", 50, 20, dict(rect=(10, 5, 80, 30), opacity=0.5)),
                                        ("img_bgra_bgra_tx", 50, 20, dict(rect=(60, 2, 60, 36)))]),
    "tex_flip":      (192, 40, True, [("img_bgra_bgra_tx", 96, 54, dict(tex=(1.0, 0.0, -1.0, 1.0))), ("img_rgba_bgra_tx", 96, 54, dict(tex=(0.2, 0.1, 0.5, 0.7), opacity=0.5))]),
    "eight_layers":  (128, 48, True, [("img_bgra_bgra_tx" if i % 2 else "img_rgba_bgra_tx", 128, 48, dict(opacity=1.0 - 0.1 * i)) for i in range(8)]),
    "odd_tiny":      (5, 3, True, [("img_bgra_bgra_tx", 7, 5, dict()), ("img_bgra_bgra_tx", 3, 3, dict(opacity=0.5))]),
    "opacity_gt_1":  (100, 30, True, [("img_bgra_bgra_tx", 100, 30, dict(opacity=1.7)), ("img_bgra_bgra_tx", 100, 30, dict(opacity=-0.3))]),
}


@pytest.mark.parametrize("case", list(RGB_CASES))
def test_rgb_layers_tiled_matches_oracle(ctx, case):
    cw, ch, clear, specs = RGB_CASES[case]
    canvas0 = util.alloc_image("bgra", cw, ch, seed=61)
    exp = util.copy_image(canvas0)
    if clear:
        assert O.run_kernel("img_clear_bgra", exp) == 0
    layers = []
    for i, (k, sw, sh, kw) in enumerate(specs):
        u = util.make_uniforms((cw, ch), in_size=(sw, sh), **kw)
        s = k.split("_")[1]
        src = util.alloc_image(s, sw, sh, seed=70 + i)
        assert O.run_kernel(k, exp, src, u, threads=4) == 0
        layers.append((sv.defaultComputeKernelFromString(k), G.to_gpu(ctx, s, sw, sh, src), u, 0))
    gd = G.to_gpu(ctx, "bgra", cw, ch, canvas0)
    h, name, keep = G.make_batch(ctx, [(gd, clear, layers)])
    if case == "odd_tiny":        # rows shorter than one 16-byte vector cannot be staged -> general kernel
        assert name == "tick_general_bgra", name
    else:
        assert name == "tick_bgra_wave", name
    G.run_batch(ctx, h)
    G.destroy_batch(h)
    G.assert_same(G.from_gpu(ctx, gd, "bgra", cw, ch), exp, f"{case} via {name}")


def test_rgb_layers_fallbacks(ctx):
    """Any number of layers per tick is fine for the wave-per-strip kernel; same bytes."""
    cw, ch = 96, 54
    src = util.alloc_image("bgra", 48, 27, seed=5)
    gs = G.to_gpu(ctx, "bgra", 48, 27, src)
    u = util.make_uniforms((cw, ch), in_size=(48, 27), opacity=0.9)
    many = [(sv.ComputeKernel.img_bgra_bgra_tx, gs, u, 0)] * 9
    exp = util.alloc_image("bgra", cw, ch)
    assert O.run_kernel("img_clear_bgra", exp) == 0
    for _ in range(9):
        assert O.run_kernel("img_bgra_bgra_tx", exp, src, u) == 0
    gd = G.to_gpu(ctx, "bgra", cw, ch, util.alloc_image("bgra", cw, ch, seed=1))
    h, name, keep = G.make_batch(ctx, [(gd, True, many)])
    assert name == "tick_bgra_wave"
    G.run_batch(ctx, h)
    G.destroy_batch(h)
    G.assert_same(G.from_gpu(ctx, gd, "bgra", cw, ch), exp, "nine layers")


# ---- y420p -> BGRA through the planar variant of the tiled kernel -------------------------------------
@pytest.mark.parametrize("case", [c for c in NV12_BGRA_CASES if c not in ("huge_downscale", "down_2.5x", "tiny")])
def test_y420p_bgra_tiled_matches_oracle(ctx, case):
    cw, ch, sw, sh, kw, clear = NV12_BGRA_CASES[case]
    u = util.make_uniforms((cw, ch), in_size=(sw, sh), **kw)
    src = util.alloc_image("y420p", sw, sh, seed=31)
    canvas0 = util.alloc_image("bgra", cw, ch, seed=32)
    exp = util.copy_image(canvas0)
    if clear:
        assert O.run_kernel("img_clear_bgra", exp) == 0
    assert O.run_kernel("img_y420p_bgra", exp, src, u, csc=1, threads=4) == 0
    gs = G.to_gpu(ctx, "y420p", sw, sh, src)
    gd = G.to_gpu(ctx, "bgra", cw, ch, canvas0)
    h, name, keep = G.make_batch(ctx, [(gd, clear, [(sv.ComputeKernel.img_y420p_bgra, gs, u, 1)])])
    assert name == "tick_y420p_bgra_tiled", name
    G.run_batch(ctx, h)
    G.destroy_batch(h)
    G.assert_same(G.from_gpu(ctx, gd, "bgra", cw, ch), exp, f"{case} via {name}")


def test_y420p_1080p_to_720p_full_size(ctx):
    sw, sh, dw, dh = 1920, 1080, 1280, 720
    src = util.alloc_image("y420p", sw, sh, seed=41)
    u = util.full_canvas_uniforms((dw, dh), (sw, sh))
    exp = util.alloc_image("bgra", dw, dh)
    assert O.run_kernel("img_clear_bgra", exp, threads=16) == 0
    assert O.run_kernel("img_y420p_bgra", exp, src, u, threads=16) == 0
    gs, gd = G.to_gpu(ctx, "y420p", sw, sh, src), G.to_gpu(ctx, "bgra", dw, dh, util.alloc_image("bgra", dw, dh, seed=3))
    h, name, keep = G.make_batch(ctx, [(gd, True, [(sv.ComputeKernel.img_y420p_bgra, gs, u, 0)])])
    assert name == "tick_y420p_bgra_tiled"
    G.run_batch(ctx, h)
    G.destroy_batch(h)
    G.assert_same(G.from_gpu(ctx, gd, "bgra", dw, dh), exp, "y420p cfg2")


# ---- 4:2:0 canvases (the reference's own kernels): batched multi-layer ticks, general quad kernel --------
YUV_CASES = {
    # name: (target fmt, canvas w, h, clear_first, [(kernel, src w, h, make_uniforms kwargs)])
    "nv12_copy":      ("nv12", 128, 48, True, [("img_nv12_nv12", 128, 48, dict())]),
    "nv12_scale":     ("nv12", 192, 108, True, [("img_nv12_nv12", 288, 162, dict(opacity=0.8))]),
    "y420p_mixer":    ("y420p", 192, 108, True, [("img_y420p_y420p", 192, 108, dict()),
                                                  ("img_bgra_y420p", 64, 36, dict(rect=(8, 8, 64, 36), opacity=0.8)),
                                                  ("img_rgba_y420p", 64, 36, dict(rect=(100, 60, 80, 44), opacity=0.6, fill=(0.2, 0.9, 0.1, 0.5)))]),
    "nv12_mixed":     ("nv12", 260, 70, True, [("img_y420p_nv12", 96, 54, dict(rect=(33, 9, 180, 40), border=(5, 3, 7, 2), fill=(0.9, 0.2, 0.1, 0.6), opacity=0.8)),
                                                ("img_bgra_nv12", 50, 40, dict(rect=(-20, -10, 120, 60), fill=(0.1, 0.5, 0.9, 1.0), opacity=0.35)),
                                                ("img_nv12_nv12", 64, 64, dict(rect=(150, 5, 90, 60), tex=(0.25, 0.0, 0.5, 1.0), fill=(1, 1, 0, 1)))]),
    "y420p_noclear":  ("y420p", 130, 38, False, [("img_y420p_y420p", 50, 20, dict(rect=(10, 5, 80, 30), opacity=0.5)),
                                                  ("img_bgra_y420p", 50, 20, dict(rect=(60, 2, 60, 36)))]),
    "nv12_flip":      ("nv12", 192, 40, True, [("img_nv12_nv12", 96, 54, dict(tex=(1.0, 0.0, -1.0, 1.0))),
                                                ("img_rgba_nv12", 96, 54, dict(tex=(0.2, 0.1, 0.5, 0.7), opacity=0.5))]),
    "y420p_tiny":     ("y420p", 4, 2, True, [("img_y420p_y420p", 6, 4, dict()), ("img_bgra_y420p", 3, 3, dict(opacity=0.5))]),
    "nv12_12_layers": ("nv12", 128, 32, True, [("img_bgra_nv12" if i % 3 else "img_nv12_nv12", 64, 16, dict(rect=(4 * i, i, 64, 16), opacity=1.0 - 0.05 * i)) for i in range(12)]),
}


@pytest.mark.parametrize("case", list(YUV_CASES))
def test_yuv_layer_ticks_match_oracle(ctx, case):
    d, cw, ch, clear, specs = YUV_CASES[case]
    canvas0 = util.alloc_image(d, cw, ch, seed=81)
    exp = util.copy_image(canvas0)
    if clear:
        assert O.run_kernel(f"img_clear_{d}", exp) == 0
    layers = []
    for i, (k, sw, sh, kw) in enumerate(specs):
        u = util.make_uniforms((cw, ch), in_size=(sw, sh), **kw)
        s = k.split("_")[1]
        src = util.alloc_image(s, sw, sh, seed=90 + i)
        assert O.run_kernel(k, exp, src, u, threads=4) == 0
        layers.append((sv.defaultComputeKernelFromString(k), G.to_gpu(ctx, s, sw, sh, src), u, 0))
    gd = G.to_gpu(ctx, d, cw, ch, canvas0)
    h, name, keep = G.make_batch(ctx, [(gd, clear, layers)])
    # sources with rows shorter than one 16-byte vector cannot be staged -> general quad kernel; everything else here is
    # axis-aligned on an even canvas -> one wave per strip (tests/test_gpu_yuvwave.py covers that kernel in depth)
    # — or, for cleared ticks of at most four layers inside the rings' limits, streamed rows (tests/test_gpu_yuvstream.py)
    assert name == f"tick_general_yuv<{d}>" if case in ("y420p_tiny",) else name in (f"tick_yuv_wave<{d}>", f"tick_yuv_stream<{d}>"), name
    G.run_batch(ctx, h)
    G.destroy_batch(h)
    G.assert_same(G.from_gpu(ctx, gd, d, cw, ch), exp, f"{case} via {name}")


def test_yuv_odd_canvas_uses_general_kernel(ctx):
    gs = G.to_gpu(ctx, "bgra", 8, 8, util.alloc_image("bgra", 8, 8, seed=1))
    gd = G.to_gpu(ctx, "nv12", 33, 17, util.alloc_image("nv12", 33, 17, seed=2))
    u = util.make_uniforms((33, 17), in_size=(8, 8))
    h, name, keep = G.make_batch(ctx, [(gd, True, [(sv.ComputeKernel.img_bgra_nv12, gs, u, 0)])])
    G.destroy_batch(h)
    assert name == "tick_general_yuv<nv12>"


def test_bounding_box_skipping_keeps_results(ctx):
    """Small overlays on a big canvas: layers are skipped per tile / per pixel by the host-side bounding box;
    edges of the box (border quads with fractional coordinates, negative scales) must not lose pixels."""
    cw, ch = 400, 120
    rng = np.random.default_rng(7)
    for d in ("bgra", "nv12"):
        exp = util.alloc_image(d, cw, ch)
        assert O.run_kernel(f"img_clear_{d}", exp) == 0
        layers = []
        for i in range(6):
            sw, sh = 40, 24
            rect = (float(rng.uniform(-30, cw - 10)), float(rng.uniform(-20, ch - 5)), float(rng.uniform(8, 120)), float(rng.uniform(6, 60)))
            kw = dict(rect=rect, border=tuple(float(v) for v in rng.uniform(0, 9, 4)), fill=(0.3, 0.6, 0.9, 0.7), opacity=float(rng.uniform(0.3, 1)))
            if i % 2:
                kw["tex"] = (1.0, 0.0, -1.0, 1.0)
            u = util.make_uniforms((cw, ch), in_size=(sw, sh), **kw)
            k = "img_bgra_bgra_tx" if d == "bgra" else "img_bgra_nv12"
            src = util.alloc_image("bgra", sw, sh, seed=200 + i)
            assert O.run_kernel(k, exp, src, u) == 0
            layers.append((sv.defaultComputeKernelFromString(k), G.to_gpu(ctx, "bgra", sw, sh, src), u, 0))
        # a rotated layer too (general kernel, whole-canvas box)
        gd = G.to_gpu(ctx, d, cw, ch, util.alloc_image(d, cw, ch, seed=3))
        h, name, keep = G.make_batch(ctx, [(gd, True, layers)])
        G.run_batch(ctx, h)
        G.destroy_batch(h)
        G.assert_same(G.from_gpu(ctx, gd, d, cw, ch), exp, f"bbox {d} via {name}")
