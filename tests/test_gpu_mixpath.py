"""tick_bgra_wave (one wave per canvas strip, no block barriers) — N layers of ANY mix of NV12 / y420p / BGRA / RGBA onto a BGRA canvas in one LDS-tiled launch
(the literal "NV12 -> BGRA + scale + 4-layer composite" tick of VideoMixer.mix, mix.video.swift:114-124) — gives exactly
the bytes of the oracle's clear + per-layer kernel calls: tile edges, scale factors, partial cover, borders, fill,
opacity, flips, odd sizes, covered-tile culling, un-cleared canvases, the staging tail and the unstaged fallback."""
import numpy as np
import pytest

import gpuutil as G
import util
from oracle import oracle as O
from swiftvideo_amd import chipvideo as cv
from swiftvideo_amd import compute as sv
from test_gpu_fastpath import NV12_BGRA_CASES, RGB_CASES

pytestmark = pytest.mark.gpu

WAVE = "tick_bgra_wave"


@pytest.fixture(params=["8", "16"])
def path(request, switch):
    """route every eligible BGRA-canvas batch through the wave kernel (also where the single-purpose kernel would be chosen),
    once with 8-row and once with 16-row strips (the host picks per launch; CHV_WAVE_ROWS forces it); yields the kernel name"""
    switch("CHV_BGRA_PATH", "wave")
    switch("CHV_WAVE_ROWS", request.param)
    return WAVE


def run_tick_case(ctx, cw, ch, clear, specs, seed=61, expect=WAVE, csc=0):
    """specs: [(kernel name, src w, h, make_uniforms kwargs)] -> asserts HIP == oracle, returns the dispatched kernel name"""
    canvas0 = util.alloc_image("bgra", cw, ch, seed=seed)
    exp = util.copy_image(canvas0)
    if clear:
        assert O.run_kernel("img_clear_bgra", exp) == 0
    layers = []
    for i, (k, sw, sh, kw) in enumerate(specs):
        kw = dict(kw)
        lcsc = kw.pop("csc", csc)               # (a layer may name its own colourspace)
        u = util.make_uniforms((cw, ch), in_size=(sw, sh), **kw)
        s = k.split("_")[1]
        src = util.alloc_image(s, sw, sh, seed=seed + 9 + i)
        assert O.run_kernel(k, exp, src, u, csc=lcsc, threads=4) == 0
        layers.append((sv.defaultComputeKernelFromString(k), G.to_gpu(ctx, s, sw, sh, src), u, lcsc))
    gd = G.to_gpu(ctx, "bgra", cw, ch, canvas0)
    h, name, keep = G.make_batch(ctx, [(gd, clear, layers)])
    if expect is not None:
        assert name == expect, f"dispatched to {name}"
    G.run_batch(ctx, h)
    G.destroy_batch(h)
    G.assert_same(G.from_gpu(ctx, gd, "bgra", cw, ch), exp, f"via {name}")
    return name


MIXED_CASES = {
    # name: (canvas w, h, clear_first, [(kernel, src w, h, make_uniforms kwargs)])
    "pipeline_small": (320, 180, True, [("img_nv12_bgra", 480, 270, dict(opacity=o)) for o in (1.0, 0.75, 0.5, 0.25)]),
    "pipeline_y420p": (320, 180, True, [("img_y420p_bgra", 480, 270, dict(opacity=o)) for o in (1.0, 0.75, 0.5, 0.25)]),
    "video_overlays": (320, 180, True, [("img_nv12_bgra", 480, 270, dict()),
                                         ("img_bgra_bgra_tx", 96, 54, dict(rect=(12, 10, 96, 54), opacity=0.8)),
                                         ("img_rgba_bgra_tx", 64, 36, dict(rect=(200, 100, 100, 60), opacity=0.6, border=(3, 3, 3, 3), fill=(0.1, 0.9, 0.2, 0.7)))]),
    "all_four_kinds": (260, 70, True, [("img_y420p_bgra", 96, 54, dict(rect=(0, 0, 130, 70))),
                                        ("img_nv12_bgra", 96, 54, dict(rect=(130, 0, 130, 70), opacity=0.9)),
                                        ("img_rgba_bgra_tx", 50, 40, dict(rect=(-20, -10, 120, 60), fill=(0.1, 0.5, 0.9, 1.0), opacity=0.35)),
                                        ("img_bgra_bgra_tx", 33, 17, dict(rect=(150, 5, 90, 60), tex=(0.25, 0.0, 0.5, 1.0), fill=(1, 1, 0, 1)))]),
    "grid_2x2":       (256, 128, True, [("img_nv12_bgra", 192, 108, dict(rect=(0, 0, 128, 64))), ("img_y420p_bgra", 192, 108, dict(rect=(128, 0, 128, 64))),
                                         ("img_nv12_bgra", 96, 54, dict(rect=(0, 64, 128, 64))), ("img_bgra_bgra_tx", 192, 108, dict(rect=(128, 64, 128, 64)))]),
    "covered_top":    (200, 70, True, [("img_bgra_bgra_tx", 200, 70, dict(opacity=0.7)), ("img_nv12_bgra", 64, 36, dict(rect=(5, 5, 60, 30), opacity=0.5)),
                                        ("img_y420p_bgra", 300, 106, dict())]),     # opaque full-canvas picture on top: everything beneath is culled
    "covered_noclear": (200, 70, False, [("img_rgba_bgra_tx", 200, 70, dict(opacity=0.7)), ("img_nv12_bgra", 300, 106, dict()),
                                          ("img_bgra_bgra_tx", 40, 20, dict(rect=(150, 40, 40, 20), opacity=0.5))]),
    "covered_partial": (200, 70, False, [("img_rgba_bgra_tx", 200, 70, dict(opacity=0.7)),
                                          ("img_nv12_bgra", 96, 54, dict(rect=(30, 0, 140, 70)))]),    # covers the middle tiles only
    "covered_fill":   (200, 70, False, [("img_nv12_bgra", 96, 54, dict(opacity=0.4)), ("img_nv12_bgra", 300, 106, dict(fill=(0.3, 0.2, 0.9, 1.0)))]),
    "noclear_blend":  (260, 70, False, [("img_nv12_bgra", 96, 54, dict(rect=(-20, -10, 200, 100), opacity=0.35, fill=(0.1, 0.5, 0.9, 1.0), border=(40, 40, 40, 40))),
                                         ("img_y420p_bgra", 64, 64, dict(rect=(150, 5, 90, 60), opacity=0.8))]),
    "flips":          (192, 40, True, [("img_nv12_bgra", 96, 54, dict(tex=(1.0, 0.0, -1.0, 1.0))), ("img_y420p_bgra", 96, 54, dict(tex=(0.2, 1.0, 0.5, -0.7), opacity=0.5))]),
    "eight_layers":   (128, 48, True, [(("img_nv12_bgra", "img_rgba_bgra_tx", "img_y420p_bgra", "img_bgra_bgra_tx")[i % 4], 128, 48, dict(opacity=1.0 - 0.1 * i)) for i in range(8)]),
    "opacity_gt_1":   (100, 30, True, [("img_nv12_bgra", 100, 30, dict(opacity=1.7)), ("img_y420p_bgra", 100, 30, dict(opacity=-0.3))]),
    "upscale_3x":     (300, 90, True, [("img_nv12_bgra", 100, 30, dict(opacity=0.5)), ("img_y420p_bgra", 100, 30, dict(opacity=0.5))]),
    "odd_width":      (131, 19, True, [("img_nv12_bgra", 98, 42, dict()), ("img_bgra_bgra_tx", 97, 41, dict(opacity=0.5))]),
    "down_2x_tail":   (160, 64, True, [("img_nv12_bgra", 320, 128, dict()), ("img_y420p_bgra", 352, 140, dict(opacity=0.5))]),   # luma rectangles > 512 slots: staging tail
    "down_4x":        (96, 40, True, [("img_nv12_bgra", 384, 160, dict(opacity=0.6)), ("img_bgra_bgra_tx", 384, 160, dict(opacity=0.5))]),
    # rectangles taller than two rows per strip row are staged as every row's OWN pair of tap rows (WGeom::pair, 8-row strips): a 2 x 2 grid of
    # 3:1 reductions of every source class over a 1.5:1 background, the same across the canvas edges and flipped, 5:1 (chroma pairs as well), a
    # vertical-only reduction (narrow rectangles: not the shift-and-mask staging), a reduction next to an enlargement
    "down_3x_grid":   (128, 72, True, [("img_nv12_bgra", 192, 108, dict()), ("img_nv12_bgra", 192, 108, dict(rect=(0, 0, 64, 36), opacity=0.9)),
                                        ("img_y420p_bgra", 192, 108, dict(rect=(64, 0, 64, 36), opacity=0.8)),
                                        ("img_bgra_bgra_tx", 192, 108, dict(rect=(0, 36, 64, 36), opacity=0.7)),
                                        ("img_rgba_bgra_tx", 192, 108, dict(rect=(64, 36, 64, 36)))]),
    "down_3x_edges":  (160, 64, True, [("img_nv12_bgra", 576, 324, dict(rect=(-20, -10, 192, 108))),
                                        ("img_y420p_bgra", 480, 200, dict(tex=(1.0, 1.0, -1.0, -1.0), opacity=0.5)),
                                        ("img_bgra_bgra_tx", 200, 180, dict(rect=(100, -8, 70, 60), opacity=0.6, border=(2, 2, 2, 2), fill=(0.2, 0.3, 0.9, 0.8)))]),
    "down_5x_yuv":    (128, 72, False, [("img_nv12_bgra", 640, 360, dict(opacity=0.8)), ("img_y420p_bgra", 640, 360, dict(opacity=0.5))]),
    "down_vertical":  (128, 72, True, [("img_nv12_bgra", 128, 216, dict()), ("img_y420p_bgra", 128, 288, dict(opacity=0.5)),
                                        ("img_rgba_bgra_tx", 40, 200, dict(rect=(60, 4, 40, 64), opacity=0.7))]),
    "down_and_up":    (192, 48, True, [("img_y420p_bgra", 64, 160, dict()), ("img_nv12_bgra", 600, 20, dict(opacity=0.5))]),
    # strips inside the inner box of an opaque YUV-source picture start at that picture (LF_COVERS): an inset over a background, opaque quadrants
    # over a background and under a translucent one, an opaque picture with a painted border ring (the ring still blends), an opaque RGB picture
    # (per-pixel alpha: nothing is culled), an uncleared canvas, an opaque picture hanging off the canvas, flipped
    "pip_opaque":     (320, 96, True, [("img_nv12_bgra", 480, 144, dict()), ("img_bgra_bgra_tx", 64, 36, dict(rect=(10, 10, 64, 36), opacity=0.7)),
                                        ("img_y420p_bgra", 240, 72, dict(rect=(70, 8, 230, 80)))]),
    "grid_opaque":    (256, 128, True, [("img_nv12_bgra", 384, 192, dict())] +
                                        [(("img_nv12_bgra", "img_y420p_bgra")[i % 2], 192, 108, dict(rect=(128 * (i % 2), 64 * (i // 2), 128, 64))) for i in range(4)] +
                                        [("img_rgba_bgra_tx", 64, 32, dict(rect=(96, 48, 64, 32), opacity=0.5))]),
    "cover_ring":     (256, 96, False, [("img_rgba_bgra_tx", 256, 96, dict(opacity=0.9)),
                                         ("img_nv12_bgra", 192, 72, dict(rect=(24, 12, 200, 72), border=(10, 6, 14, 8), fill=(0.2, 0.8, 0.3, 0.6)))]),
    "cover_rgb_not":  (192, 64, True, [("img_nv12_bgra", 192, 64, dict()), ("img_bgra_bgra_tx", 192, 64, dict())]),
    "cover_offcanvas": (200, 80, False, [("img_y420p_bgra", 100, 40, dict(opacity=0.6)), ("img_nv12_bgra", 300, 120, dict(rect=(-40, -20, 300, 120))),
                                          ("img_y420p_bgra", 96, 54, dict(rect=(60, 10, 130, 60), tex=(1.0, 1.0, -1.0, -1.0)))]),
}


# LF_SAME_GEOM (device_types.h): a layer whose three matrices, source plane sizes / pitches / class and bounding box equal its
# predecessor's keeps the predecessor's column entry, row table and rectangles — only opacity, fill colour, colourspace and the
# plane pointers differ.  Ticks that interleave such layers with layers of other geometry, source class or size:
A = dict()
R = dict(rect=(40, 20, 200, 120))
OFF = dict(rect=(-30, -12, 300, 170))           # crosses every canvas edge: edge staging + masked rows
BF = dict(rect=(50, 30, 180, 100), border=(6, 4, 6, 4), fill=(0.9, 0.2, 0.1, 0.6))
SAME_GEOM_CASES = {
    "interleaved": (320, 180, True, [("img_nv12_bgra", 480, 270, dict(A)), ("img_nv12_bgra", 480, 270, dict(A, opacity=0.75)),
                                     ("img_nv12_bgra", 480, 270, dict(R, opacity=0.9)), ("img_nv12_bgra", 480, 270, dict(R, opacity=0.5)),
                                     ("img_y420p_bgra", 480, 270, dict(R, opacity=0.5)), ("img_y420p_bgra", 480, 270, dict(R, opacity=0.25)),
                                     ("img_nv12_bgra", 480, 270, dict(A, opacity=0.3)), ("img_nv12_bgra", 240, 134, dict(A, opacity=0.3)),
                                     ("img_bgra_bgra_tx", 96, 54, dict(R, opacity=0.8)), ("img_bgra_bgra_tx", 96, 54, dict(R, opacity=0.6)),
                                     ("img_rgba_bgra_tx", 96, 54, dict(R, opacity=0.6))]),
    "off_canvas": (320, 180, False, [("img_nv12_bgra", 300, 170, dict(OFF, opacity=0.8)), ("img_nv12_bgra", 300, 170, dict(OFF, opacity=0.6)),
                                     ("img_y420p_bgra", 300, 170, dict(OFF, opacity=0.6)), ("img_y420p_bgra", 300, 170, dict(OFF))]),
    "border_fill": (320, 180, True, [("img_nv12_bgra", 200, 110, dict(BF, opacity=0.8)), ("img_nv12_bgra", 200, 110, dict(BF, opacity=0.5)),
                                     ("img_nv12_bgra", 200, 110, dict(BF, opacity=0.5, fill=(0.1, 0.3, 0.9, 0.0))),
                                     ("img_bgra_bgra_tx", 200, 110, dict(BF, opacity=0.5)), ("img_bgra_bgra_tx", 200, 110, dict(BF, opacity=1.0))]),
    "native_rgb": (256, 96, True, [("img_bgra_bgra_tx", 256, 96, dict(opacity=o)) for o in (1.0, 0.75, 0.5, 0.25)]),
    "csc_differs": (192, 64, True, [("img_nv12_bgra", 288, 96, dict()), ("img_nv12_bgra", 288, 96, dict(opacity=0.5))]),
}


@pytest.mark.parametrize("case", list(SAME_GEOM_CASES))
@pytest.mark.parametrize("share", ["1", "0"])
def test_layers_sharing_geometry_match_oracle(ctx, path, switch, case, share):
    """same- and different-geometry layers interleaved in one tick, both strip heights, with and without the sharing"""
    switch("CHV_SAME_GEOM", share)
    cw, ch, clear, specs = SAME_GEOM_CASES[case]
    run_tick_case(ctx, cw, ch, clear, specs, expect=path, seed=131)


def _stack(kernel, sw, sh, base, ops):
    return [(kernel, sw, sh, dict(base, opacity=o)) for o in ops]


PAIR_CASES = {
    # launches of ONE YUV source class (the single-class instantiations of the kernel), same-geometry runs of 2-5 layers
    "pipeline_small":  (320, 180, True, _stack("img_nv12_bgra", 480, 270, A, (1.0, 0.75, 0.5, 0.25))),
    "translucent_first_uncleared": (320, 180, False, _stack("img_nv12_bgra", 480, 270, A, (0.6, 0.3))),
    "three_and_five":  (320, 180, True, _stack("img_nv12_bgra", 480, 270, A, (1.0, 0.5, 0.25)) + _stack("img_nv12_bgra", 200, 110, R, (0.9, 0.7, 0.5, 0.3, 0.1))),
    "edges_masked":    (320, 180, False, _stack("img_nv12_bgra", 300, 170, OFF, (0.8, 0.6, 1.0, 0.4))),
    "planar":          (320, 180, True, _stack("img_y420p_bgra", 480, 270, A, (1.0, 0.5)) + _stack("img_y420p_bgra", 200, 110, R, (0.5, 1.0, 0.25))),
    "planar_edges":    (260, 70, True, _stack("img_y420p_bgra", 300, 170, OFF, (1.0, 0.35))),
    "opaque_on_top":   (192, 64, True, _stack("img_nv12_bgra", 288, 96, A, (0.5, 1.0, 1.0, 0.5))),
    "upscaled":        (300, 90, True, _stack("img_nv12_bgra", 100, 30, A, (1.0, 0.5, 0.5))),
    "own_colourspaces": (192, 64, True, [("img_nv12_bgra", 288, 96, dict(A, csc=c, opacity=o)) for c, o in ((0, 1.0), (1, 0.5), (2, 0.5), (3, 0.25))]),
    # fill paint and opacities outside [0, 1] between layers of one geometry
    "unpairable":      (320, 180, True, [("img_nv12_bgra", 200, 110, dict(BF, opacity=0.8)), ("img_nv12_bgra", 200, 110, dict(BF, opacity=0.5)),
                                         ("img_nv12_bgra", 480, 270, dict(A, opacity=1.5)), ("img_nv12_bgra", 480, 270, dict(A, opacity=0.5)),
                                         ("img_nv12_bgra", 480, 270, dict(A, opacity=-0.25)), ("img_nv12_bgra", 480, 270, dict(A, opacity=0.5)),
                                         ("img_nv12_bgra", 480, 270, dict(A, opacity=0.5, fill=(0.2, 0.4, 0.6, 0.5)))]),
}


@pytest.mark.parametrize("case", list(PAIR_CASES))
def test_single_class_stacks_match_oracle(ctx, path, case):
    """stacks of same-geometry layers of ONE YUV class (the shape of the headline tick), per-layer colourspaces, opaque layers on top,
    layers that paint fill or have opacities outside [0, 1] in between"""
    cw, ch, clear, specs = PAIR_CASES[case]
    run_tick_case(ctx, cw, ch, clear, specs, expect=path, seed=151)


STREAM = "tick_bgra_stream"
STREAM_CASES = {
    # ticks of 1..4 full-frame NV12 layers of ONE geometry on a cleared canvas: rows outermost, layers innermost (kernels_stream.hip.cpp);
    # plane rows are multiples of 16 bytes, the horizontal reduction is at most 1.7
    "pipeline_small":   (320, 180, _stack("img_nv12_bgra", 480, 272, A, (1.0, 0.75, 0.5, 0.25))),
    "two_layers":       (320, 180, _stack("img_nv12_bgra", 480, 272, A, (0.6, 0.3))),
    "three_opaque_top": (192, 64, _stack("img_nv12_bgra", 288, 96, A, (0.5, 1.0, 1.0))),
    "one_layer":        (320, 180, _stack("img_nv12_bgra", 480, 272, A, (1.0,))),
    "native_size":      (320, 192, _stack("img_nv12_bgra", 320, 192, A, (1.0, 0.5, 0.5, 0.5))),
    "enlarged":         (322, 182, _stack("img_nv12_bgra", 160, 96, A, (1.0, 0.5))),
    "inside_canvas":    (320, 180, _stack("img_nv12_bgra", 304, 176, R, (0.9, 0.7, 0.5, 0.3))),
    "across_edges":     (320, 180, _stack("img_nv12_bgra", 304, 176, OFF, (0.8, 0.6, 1.0, 0.4))),
    "tall":             (192, 700, _stack("img_nv12_bgra", 288, 1056, A, (1.0, 0.5, 0.25))),
    "wide":             (1280, 40, _stack("img_nv12_bgra", 1920, 64, A, (1.0, 0.75, 0.5, 0.25))),
    "own_colourspaces": (192, 64, [("img_nv12_bgra", 288, 96, dict(A, csc=c, opacity=o)) for c, o in ((0, 1.0), (1, 0.5), (2, 0.5), (3, 0.25))]),
    "tex_window":       (320, 180, _stack("img_nv12_bgra", 480, 272, dict(tex=(0.05, 0.1, 0.95, 0.9)), (1.0, 0.5))),
}
# the same ticks with PLANAR sources (y420p: what FFmpeg's software decoders emit): a third ring per layer; chroma rows are multiples of 16 bytes
# where the source width is a multiple of 32
STREAM_CASES.update({name + "_y420p": (cw, ch, [(k.replace("nv12", "y420p"), sw, sh, kw) for k, sw, sh, kw in specs])
                     for name, (cw, ch, specs) in list(STREAM_CASES.items()) if all(sw % 32 == 0 for _, sw, _, _ in specs)})


@pytest.mark.parametrize("case", list(STREAM_CASES))
def test_stream_kernel_matches_oracle(ctx, switch, case):
    switch("CHV_BGRA_PATH", "stream")
    cw, ch, specs = STREAM_CASES[case]
    run_tick_case(ctx, cw, ch, True, specs, expect=STREAM, seed=171)


@pytest.mark.parametrize("desc", ["arguments", "device", "host"])
@pytest.mark.parametrize("case", list(STREAM_CASES))
def test_lone_stream_tick_matches_oracle(ctx, switch, case, desc):
    """One tick at a time (chv_composite, what a Swift VideoMixer issues): the streaming kernel takes the tick's descriptors as kernel
    arguments (tick_bgra_stream_one); CHV_DESC=device / host: the copy in device memory / the pinned host ring, through the pointer kernel."""
    if desc != "arguments":
        switch("CHV_DESC", desc)
    cw, ch, specs = STREAM_CASES[case]
    if len(specs) == 1 and desc != "arguments":
        switch("CHV_BGRA_PATH", "stream")                # (a lone one-layer tick takes the streaming kernel only with by-value descriptors)
    canvas0 = util.alloc_image("bgra", cw, ch, seed=333)
    exp = util.copy_image(canvas0)
    assert O.run_kernel("img_clear_bgra", exp) == 0
    layers = []
    for i, (k, sw, sh, kw) in enumerate(specs):
        kw = dict(kw)
        lcsc = kw.pop("csc", 0)
        u = util.make_uniforms((cw, ch), in_size=(sw, sh), **kw)
        fmt = k.split("_")[1]
        src = util.alloc_image(fmt, sw, sh, seed=340 + i)
        assert O.run_kernel(k, exp, src, u, csc=lcsc, threads=4) == 0
        layers.append((sv.defaultComputeKernelFromString(k), G.to_gpu(ctx, fmt, sw, sh, src), u, lcsc))
    gd = G.to_gpu(ctx, "bgra", cw, ch, canvas0)
    for _ in range(2):                                   # (twice: the second tick finds the first one's canvas)
        sv.usingContext(ctx, lambda c: sv.compositeTick(c, gd, layers, True))
    G.assert_same(G.from_gpu(ctx, gd, "bgra", cw, ch), exp, f"lone tick, descriptors: {desc}")


def _random_stream_ticks(ctx, seed, nl_low=2):
    """three ticks of different canvas sizes, nl_low..4 NV12 layers of one geometry each -> (ticks, expected canvases, (canvas, w, h))"""
    rng = np.random.default_rng(7100 + seed)
    nl = int(rng.integers(nl_low, 5))
    fmt = "y420p" if seed % 3 == 2 else "nv12"         # (planar sources: a launch's layers are all of one class)
    unit = 32 if fmt == "y420p" else 16
    ticks, exps, gds = [], [], []
    for t in range(3):
        cw, ch = int(rng.integers(20, 330)) * 2, int(rng.integers(8, 200)) * 2
        sw, sh = int(rng.integers(2, 640 // unit)) * unit, int(rng.integers(4, 150)) * 2
        kw = {}
        if rng.random() < 0.6:
            rw = float(rng.uniform(0.62, 3.0)) * sw                 # the picture's width on the canvas: source texels per pixel <= 1.6
            kw["rect"] = (float(rng.uniform(-0.3, 0.5) * cw), float(rng.uniform(-0.3, 0.5) * ch), rw, float(rng.uniform(0.2, 2.5)) * sh)
        else:
            if sw / cw > 1.6:
                sw = max(unit, int(cw * 1.5) // unit * unit)
        if rng.random() < 0.25:
            kw["tex"] = (0.0, 0.0, float(rng.uniform(1.0, 1.5)), float(rng.uniform(0.6, 1.4)))
        canvas0 = util.alloc_image("bgra", cw, ch, seed=int(rng.integers(1, 1 << 20)))
        exp = util.copy_image(canvas0)
        assert O.run_kernel("img_clear_bgra", exp) == 0
        layers = []
        for l in range(nl):
            op = float(rng.choice([1.0, rng.uniform(0, 1)]))
            csc = int(rng.integers(0, 4))
            u = util.make_uniforms((cw, ch), in_size=(sw, sh), opacity=op, **kw)
            src = util.alloc_image(fmt, sw, sh, seed=int(rng.integers(1, 1 << 20)))
            assert O.run_kernel(f"img_{fmt}_bgra", exp, src, u, csc=csc, threads=4) == 0
            layers.append((sv.defaultComputeKernelFromString(f"img_{fmt}_bgra"), G.to_gpu(ctx, fmt, sw, sh, src), u, csc))
        gd = G.to_gpu(ctx, "bgra", cw, ch, canvas0)
        ticks.append((gd, True, layers)); exps.append(exp); gds.append((gd, cw, ch))
    return ticks, exps, gds


@pytest.mark.parametrize("seed", range(36))
def test_random_stream_ticks(ctx, switch, seed):
    """Seeded random ticks of the streaming kernel's class: three ticks of different canvas sizes per launch, 2..4 NV12 layers of one
    geometry each — full canvas or a rectangle anywhere (also across the canvas edges), enlargements and reductions up to 1.7 across
    and anything down, a texture window now and then, per-layer colourspaces and opacities."""
    switch("CHV_BGRA_PATH", "stream")
    ticks, exps, gds = _random_stream_ticks(ctx, seed)
    h, name, keep = G.make_batch(ctx, ticks)
    assert name in (STREAM, WAVE), name                  # (a rectangle wider than 1.7 source texels per pixel: the strip kernel)
    G.run_batch(ctx, h)
    G.destroy_batch(h)
    for i, ((gd, cw, ch), exp) in enumerate(zip(gds, exps)):
        G.assert_same(G.from_gpu(ctx, gd, "bgra", cw, ch), exp, f"seed {seed} tick {i} via {name}")


@pytest.mark.parametrize("seed", range(100, 118))
def test_random_lone_stream_ticks(ctx, seed):
    """the same random ticks (1..4 layers), one at a time through chv_composite: descriptors as kernel arguments (tick_bgra_stream_one)"""
    ticks, exps, gds = _random_stream_ticks(ctx, seed, nl_low=1)
    for (gd, clear, layers), exp, (_, cw, ch) in zip(ticks, exps, gds):
        sv.usingContext(ctx, lambda c: sv.compositeTick(c, gd, layers, True))
        G.assert_same(G.from_gpu(ctx, gd, "bgra", cw, ch), exp, f"seed {seed}, lone tick of {len(layers)} layer(s)")


@pytest.mark.parametrize("desc", ["device", "host"])
def test_descriptor_ring_wraps_under_back_to_back_launches(ctx, switch, desc):
    """200 transient launches without a host wait in between, each with its own descriptors (a 16 x 16 picture somewhere else on the
    canvas), through a kernel that reads them from the ring (the strip kernel; the ring has 64 slots guarded by one event per eight):
    a slot rewritten before its launch had read it would misplace a picture."""
    switch("CHV_DESC", desc)
    cw, ch, n = 256, 64, 200
    canvas0 = util.alloc_image("bgra", cw, ch, seed=901)
    exp = util.copy_image(canvas0)
    srcs = [util.alloc_image("bgra", 16, 16, seed=910 + i) for i in range(5)]
    gsrc = [G.to_gpu(ctx, "bgra", 16, 16, s) for s in srcs]
    k = sv.defaultComputeKernelFromString("img_bgra_bgra_tx")
    layers = []
    for i in range(n):
        u = util.make_uniforms((cw, ch), rect=((i * 37) % 236 + 0.5 * (i % 2), (i * 11) % 46, 16, 16), opacity=0.5 + 0.5 * ((i * 7) % 10) / 10.0, in_size=(16, 16))
        assert O.run_kernel("img_bgra_bgra_tx", exp, srcs[i % 5], u) == 0
        layers.append((k, gsrc[i % 5], u, 0))
    gd = G.to_gpu(ctx, "bgra", cw, ch, canvas0)

    def burst(c):
        for l in layers:
            sv.compositeTick(c, gd, [l], False)
        return c
    sv.usingContext(ctx, burst)
    G.assert_same(G.from_gpu(ctx, gd, "bgra", cw, ch), exp, f"200 back-to-back transient launches, descriptors: {desc}")


def test_stream_kernel_eligibility(ctx, switch):
    """what it must leave to the other kernels: an un-cleared canvas, a layer of another geometry or class, fill paint, opacity
    outside [0, 1], a flip, a strong reduction, plane rows that are not a multiple of 16 bytes, more than four layers"""
    base = _stack("img_nv12_bgra", 480, 272, A, (1.0, 0.5))
    assert run_tick_case(ctx, 320, 180, True, base, expect=None) == STREAM                  # launches of every size (short chunks for small ones)
    assert run_tick_case(ctx, 320, 180, True, base[:1], expect=None) != STREAM              # one-layer ticks in a batch: only on request
    switch("CHV_BGRA_PATH", "stream")
    assert run_tick_case(ctx, 320, 180, True, base, expect=None) == STREAM
    assert run_tick_case(ctx, 320, 180, True, base[:1], expect=None) == STREAM
    assert run_tick_case(ctx, 320, 180, False, base, expect=None) != STREAM
    for bad in ([("img_nv12_bgra", 480, 272, dict(A)), ("img_nv12_bgra", 480, 272, dict(R, opacity=0.5))],
                [("img_nv12_bgra", 480, 272, dict(A)), ("img_y420p_bgra", 480, 272, dict(A, opacity=0.5))],
                _stack("img_nv12_bgra", 480, 272, dict(A, fill=(0.1, 0.2, 0.3, 0.5)), (1.0, 0.5)),
                _stack("img_nv12_bgra", 480, 272, A, (1.0, 1.5)),
                _stack("img_nv12_bgra", 480, 272, dict(tex=(1.0, 0.0, -1.0, 1.0)), (1.0, 0.5)),
                _stack("img_nv12_bgra", 640, 360, dict(A), (1.0, 0.5)),                      # 2:1 reduction onto 320 columns
                _stack("img_nv12_bgra", 300, 170, A, (1.0, 0.5)),
                # a picture squeezed into a few canvas rows (a zoom animation's first frames): the rings would be advanced through dozens
                # of source rows per canvas row, the strip kernels cull by bounding box instead — and the bytes still match the oracle
                _stack("img_nv12_bgra", 480, 272, dict(rect=(0, 60, 320, 6)), (1.0, 0.5)),
                _stack("img_nv12_bgra", 480, 272, dict(rect=(0, 90, 320, 0.01)), (1.0, 0.5)),
                _stack("img_nv12_bgra", 480, 272, A, (1.0, 0.5, 0.5, 0.5, 0.5))):
        assert run_tick_case(ctx, 320, 180, True, bad, expect=None) != STREAM
    switch("CHV_STREAM", "0")
    assert run_tick_case(ctx, 320, 180, True, base, expect=None) != STREAM
    switch("CHV_STREAM", "1")
    switch("CHV_BGRA_PATH", "")
    # the default route of a launch that fills the chip: 64 two-layer ticks
    u = [util.make_uniforms((320, 192), in_size=(480, 288), opacity=o) for o in (1.0, 0.5)]
    srcs = [util.alloc_image("nv12", 480, 288, seed=50 + i) for i in range(2)]
    exp = util.alloc_image("bgra", 320, 192)
    assert O.run_kernel("img_clear_bgra", exp) == 0
    for s_, u_ in zip(srcs, u):
        assert O.run_kernel("img_nv12_bgra", exp, s_, u_) == 0
    gs = [G.to_gpu(ctx, "nv12", 480, 288, s_) for s_ in srcs]
    gds = [G.to_gpu(ctx, "bgra", 320, 192, util.alloc_image("bgra", 320, 192, seed=7)) for _ in range(80)]
    h, name, keep = G.make_batch(ctx, [(gd, True, [(sv.ComputeKernel.img_nv12_bgra, g, uu, 0) for g, uu in zip(gs, u)]) for gd in gds])
    assert name == STREAM
    G.run_batch(ctx, h)
    G.destroy_batch(h)
    for gd in (gds[0], gds[41], gds[79]):
        G.assert_same(G.from_gpu(ctx, gd, "bgra", 320, 192), exp, "80-tick launch through the streaming kernel")


@pytest.mark.parametrize("case", list(MIXED_CASES))
@pytest.mark.parametrize("csc", [0, 1])
def test_mixed_layers_match_oracle(ctx, path, case, csc):
    cw, ch, clear, specs = MIXED_CASES[case]
    # down_4x: the 4-byte texel rectangle of a strip exceeds the LDS budget -> general kernel
    run_tick_case(ctx, cw, ch, clear, specs, csc=csc, expect=None if case == "down_4x" else path)


def test_default_route_of_a_mixed_tick(ctx, switch):
    """without any switch: a stack of NV12 videos of one geometry on a cleared canvas -> the streaming kernel (the strip kernel with
    CHV_STREAM=0); mixed kinds -> the strip kernel; one YUV layer, RGB layers only -> single-purpose kernels"""
    cw, ch, clear, specs = MIXED_CASES["pipeline_small"]
    assert run_tick_case(ctx, cw, ch, clear, specs, expect=None) == STREAM
    switch("CHV_STREAM", "0")
    assert run_tick_case(ctx, cw, ch, clear, specs, expect=None) == WAVE
    switch("CHV_STREAM", "1")
    cw, ch, clear, specs = MIXED_CASES["video_overlays"]
    assert run_tick_case(ctx, cw, ch, clear, specs, expect=None) == WAVE


# A batch whose ticks are "2..4 full-frame videos of one geometry, then something else": the streaming kernel composes the videos on the cleared
# canvas, a second launch continues on it with the rest (chv_batch_create: split_stream_prefix).  Same bytes as one pass — the canvas is
# re-quantised between layers either way.
V = dict()
SPLIT_CASES = {
    "videos_logo":     (320, 180, [("img_nv12_bgra", 480, 270, dict(opacity=o)) for o in (1.0, 0.75, 0.5, 0.25)] +
                                   [("img_rgba_bgra_tx", 80, 44, dict(rect=(200, 20, 80, 44), rotation=0.3, opacity=0.9))], 4),
    "planar_overlays": (320, 180, [("img_y420p_bgra", 480, 270, dict(opacity=o)) for o in (1.0, 0.6, 0.3)] +
                                   [("img_bgra_bgra_tx", 96, 54, dict(rect=(12, 10, 96, 54), opacity=0.8)),
                                    ("img_rgba_bgra_tx", 64, 36, dict(rect=(200, 100, 100, 60), opacity=0.6, border=(3, 3, 3, 3), fill=(0.1, 0.9, 0.2, 0.7)))], 3),
    "two_and_pip":     (256, 96, [("img_nv12_bgra", 384, 144, dict()), ("img_nv12_bgra", 384, 144, dict(opacity=0.5)),
                                  ("img_nv12_bgra", 192, 72, dict(rect=(150, 8, 96, 36)))], 2),
    "five_videos":     (192, 64, [("img_nv12_bgra", 288, 96, dict(opacity=1.0 - 0.15 * i)) for i in range(5)], 4),
    "reference_bgra":  (192, 64, [("img_nv12_bgra", 288, 96, dict(opacity=o)) for o in (1.0, 0.5)] + [("img_bgra_bgra", 64, 24, dict(rect=(10, 10, 64, 24)))], 2),
    "seven_layers":    (128, 48, [("img_y420p_bgra", 128, 48, dict(opacity=0.9))] * 4 + [("img_nv12_bgra", 128, 48, dict(opacity=0.4)), ("img_bgra_bgra_tx", 128, 48, dict(opacity=0.3)),
                                  ("img_rgba_bgra_tx", 40, 20, dict(rect=(60, 20, 40, 20)))], 4),
}


@pytest.mark.parametrize("case", list(SPLIT_CASES))
def test_batches_split_between_the_streaming_kernel_and_the_rest(ctx, case):
    cw, ch, specs, k = SPLIT_CASES[case]
    import ctypes as C
    lib = cv.load()
    ticks, exps, gds = [], [], []
    for t in range(3):                                         # three ticks of the same stack, different pictures
        canvas0 = util.alloc_image("bgra", cw, ch, seed=70 + t)
        exp = util.copy_image(canvas0)
        assert O.run_kernel("img_clear_bgra", exp) == 0
        layers = []
        for i, (kn, sw, sh, kw) in enumerate(specs):
            u = util.make_uniforms((cw, ch), in_size=(sw, sh), **kw)
            s = kn.split("_")[1]
            src = util.alloc_image(s, sw, sh, seed=300 + 17 * t + i)
            assert O.run_kernel(kn, exp, src, u, threads=4) == 0
            layers.append((sv.defaultComputeKernelFromString(kn), G.to_gpu(ctx, s, sw, sh, src), u, 0))
        gd = G.to_gpu(ctx, "bgra", cw, ch, canvas0)
        ticks.append((gd, True, layers)); exps.append(exp); gds.append(gd)
    h, name, keep = G.make_batch(ctx, ticks)
    n = C.c_int(0)
    cv.check(lib.chv_batch_describe(h, None, 0, C.byref(n)))
    assert name.startswith(STREAM + " + ") and n.value == 2, (name, n.value)
    G.run_batch(ctx, h)
    G.run_batch(ctx, h)                                        # replay: the first launch clears, so the result is the same
    G.destroy_batch(h)
    for t in range(3):
        G.assert_same(G.from_gpu(ctx, gds[t], "bgra", cw, ch), exps[t], f"{case} tick {t} via {name}")


def test_batches_that_are_not_split(ctx, switch):
    """no split: an un-cleared canvas, ticks of different depth of the video stack, a stack of one, every layer taken by the streaming kernel,
    the strip kernel asked for by switch"""
    vid = lambda o: ("img_nv12_bgra", 288, 96, dict(opacity=o))
    logo = ("img_rgba_bgra_tx", 40, 20, dict(rect=(60, 20, 40, 20)))
    assert run_tick_case(ctx, 192, 64, False, [vid(1.0), vid(0.5), logo], expect=None) == WAVE
    assert run_tick_case(ctx, 192, 64, True, [vid(1.0), logo], expect=None) == WAVE
    assert run_tick_case(ctx, 192, 64, True, [vid(1.0), vid(0.5)], expect=None) == STREAM
    switch("CHV_BGRA_PATH", "wave")
    assert run_tick_case(ctx, 192, 64, True, [vid(1.0), vid(0.5), logo], expect=None) == WAVE


@pytest.mark.parametrize("target", ["bgra", "nv12"])
@pytest.mark.parametrize("seed", range(12))
def test_bounding_boxes_of_rotated_and_sheared_layers(ctx, switch, seed, target):
    """A layer that is not axis-aligned is culled by the box of the parallelogram its border coordinates map to [0,1]^2 (chipvideo.cpp::layer_bbox):
    random rotations (any angle), anisotropic sizes, a shear multiplied into the matrices by hand, borders with fill paint, layers hanging off
    every canvas edge and layers entirely outside — through the strip kernel (per-pixel layers in the strips the box touches) and through the
    general kernel (pixels outside the box skipped), three layers per tick so that a box too small for one layer shows"""
    rng = np.random.default_rng(31000 + seed)
    cw, ch = int(rng.integers(40, 200)) * 2, int(rng.integers(20, 90)) * 2
    fam = ["img_bgra_bgra_tx", "img_rgba_bgra_tx", "img_nv12_bgra", "img_y420p_bgra"] if target == "bgra" else ["img_bgra_nv12", "img_rgba_nv12", "img_nv12_nv12", "img_y420p_nv12"]
    canvas0 = util.alloc_image(target, cw, ch, seed=seed + 1)
    exp = util.copy_image(canvas0)
    assert O.run_kernel(f"img_clear_{target}", exp) == 0
    layers = []
    for l in range(3):
        k = fam[int(rng.integers(0, 4))]
        s = k.split("_")[1]
        sw, sh = int(rng.integers(8, 60)) * 2, int(rng.integers(4, 40)) * 2
        w, h = float(rng.uniform(0.1, 0.9) * cw), float(rng.uniform(0.1, 0.9) * ch)
        x, y = float(rng.uniform(-0.4, 1.1) * cw), float(rng.uniform(-0.4, 1.1) * ch)
        kw = dict(rect=(x, y, w, h), rotation=float(rng.uniform(-3.2, 3.2)), opacity=float(rng.choice([1.0, rng.uniform(0.2, 1.0)])))
        if rng.random() < 0.5:
            kw["border"] = tuple(float(v) for v in rng.uniform(0, 12, 4)); kw["fill"] = tuple(float(v) for v in rng.uniform(0, 1, 4))
        u = util.make_uniforms((cw, ch), in_size=(sw, sh), **kw)
        if rng.random() < 0.5:
            # a shear on top: kernel rows are rows of M^-1, so M' = M S  <=>  rows' = S^-1 rows (transform and border alike)
            S = np.eye(4, dtype=np.float64); S[0, 1] = float(rng.uniform(-0.7, 0.7)); Si = np.linalg.inv(S)
            for o in (0, 32):
                u[o:o + 16] = (Si @ u[o:o + 16].astype(np.float64).reshape(4, 4)).astype(np.float32).reshape(-1)
        src = util.alloc_image(s, sw, sh, seed=int(rng.integers(1, 1 << 20)))
        assert O.run_kernel(k, exp, src, u, threads=4) == 0
        layers.append((sv.defaultComputeKernelFromString(k), G.to_gpu(ctx, s, sw, sh, src), u, 0))
    for general in ("0", "1"):
        switch("CHV_FORCE_GENERAL", general)
        gd = G.to_gpu(ctx, target, cw, ch, canvas0)
        h, name, keep = G.make_batch(ctx, [(gd, True, layers)])
        G.run_batch(ctx, h)
        G.destroy_batch(h)
        G.assert_same(G.from_gpu(ctx, gd, target, cw, ch), exp, f"seed {seed} {cw}x{ch} via {name}")


@pytest.mark.parametrize("case", [c for c in NV12_BGRA_CASES if c not in ("huge_downscale", "tiny")])
@pytest.mark.parametrize("fmt", ["nv12", "y420p"])
def test_single_yuv_layer_through_the_multi_layer_kernels(ctx, path, case, fmt):
    """the cases of the single-purpose NV12 -> BGRA kernel, routed through the multi-layer kernels (CHV_BGRA_PATH)"""
    cw, ch, sw, sh, kw, clear = NV12_BGRA_CASES[case]
    run_tick_case(ctx, cw, ch, clear, [(f"img_{fmt}_bgra", sw, sh, kw)], seed=21, csc=3, expect=path)


@pytest.mark.parametrize("case", [c for c in RGB_CASES if c != "odd_tiny"])
def test_rgb_cases_through_the_multi_layer_kernels(ctx, path, case):
    cw, ch, clear, specs = RGB_CASES[case]
    run_tick_case(ctx, cw, ch, clear, specs, expect=path)


def test_mixed_fallbacks(ctx, monkeypatch):
    """any number of layers per tick is fine for the wave kernel; a rotated layer in the tick -> general kernel; same bytes"""
    nine = [("img_nv12_bgra" if i % 2 else "img_bgra_bgra_tx", 48, 28, dict(opacity=0.9)) for i in range(9)]
    assert run_tick_case(ctx, 96, 54, True, nine, expect=None) == WAVE
    twenty = [(("img_nv12_bgra", "img_rgba_bgra_tx", "img_y420p_bgra", "img_bgra_bgra_tx")[i % 4], 64, 36,
               dict(rect=(3 * i, i, 64, 36), opacity=0.95 - 0.04 * i)) for i in range(20)]
    assert run_tick_case(ctx, 140, 60, True, twenty, expect=None) == WAVE
    # a rotated layer among axis-aligned ones stays in the wave kernel (applied per pixel with the general kernel's code, in z order)
    rot = [("img_nv12_bgra", 48, 28, dict()), ("img_bgra_bgra_tx", 48, 28, dict(rect=(10, 5, 40, 20), rotation=0.3))]
    assert run_tick_case(ctx, 96, 54, True, rot, expect=None) == WAVE
    # finite but unbounded matrix entries (a picture 1e-18 pixels wide: inverse entries beyond 2^60): the strip machinery evaluates the
    # short form of the geometry only, whose zero terms could then meet an overflowed product -> that layer per pixel as well
    thin = [("img_nv12_bgra", 48, 28, dict()), ("img_bgra_bgra_tx", 48, 28, dict(rect=(10, 5, 1e-18, 20), border=(3, 3, 3, 3), fill=(0.2, 0.4, 0.9, 0.8)))]
    assert run_tick_case(ctx, 96, 54, True, thin, expect=None) == WAVE
    # nothing but such layers: the general kernel
    only = [("img_bgra_bgra_tx", 48, 28, dict(rect=(10, 5, 40, 20), rotation=0.3)), ("img_nv12_bgra", 48, 28, dict(rotation=-0.1, opacity=0.5))]
    assert run_tick_case(ctx, 96, 54, True, only, expect=None) == "tick_general_bgra"


ROTATED_CASES = {
    "logo_over_videos": (320, 180, True, [("img_nv12_bgra", 480, 270, dict()), ("img_y420p_bgra", 480, 270, dict(opacity=0.5)),
                                          ("img_rgba_bgra_tx", 64, 36, dict(rect=(200, 100, 100, 60), rotation=0.4, opacity=0.8, border=(3, 3, 3, 3), fill=(0.1, 0.9, 0.2, 0.7)))]),
    "rotated_video_first": (260, 70, False, [("img_nv12_bgra", 96, 54, dict(rect=(20, 5, 200, 60), rotation=-0.2)),
                                             ("img_bgra_bgra_tx", 64, 36, dict(rect=(40, 10, 64, 36), opacity=0.6)),
                                             ("img_y420p_bgra", 96, 54, dict(rect=(100, 0, 150, 70), rotation=0.7, opacity=0.9, fill=(0.3, 0.3, 0.9, 0.5)))]),
    "every_other": (200, 120, True, [(("img_nv12_bgra", "img_bgra_bgra_tx")[i % 2], 64, 36,
                                      dict(rect=(12 * i, 8 * i, 90, 50), rotation=(0.15 * i if i % 2 else 0.0), opacity=0.9 - 0.05 * i)) for i in range(7)]),
}


@pytest.mark.parametrize("case", list(ROTATED_CASES))
def test_rotated_layers_inside_wave_ticks(ctx, path, case):
    cw, ch, clear, specs = ROTATED_CASES[case]
    assert run_tick_case(ctx, cw, ch, clear, specs, expect=None) == WAVE


METAL_CASES = {
    # what an UNCHANGED VideoMixer.findKernel issues on a BGRA canvas (mix.video.swift:167-182): video layers through img_<fmt>_bgra and
    # every BGRA layer through img_bgra_bgra (kernels.metal:52-62: nearest, whole canvas, per-pixel source alpha, no transform / opacity)
    "overlay_over_videos": (320, 180, True, [("img_nv12_bgra", 480, 270, dict()), ("img_y420p_bgra", 480, 270, dict(opacity=0.5)),
                                             ("img_bgra_bgra", 64, 36, dict())]),
    "overlay_between": (260, 70, False, [("img_nv12_bgra", 96, 54, dict(rect=(20, 5, 200, 60))), ("img_bgra_bgra", 260, 70, dict()),
                                         ("img_y420p_bgra", 96, 54, dict(rect=(100, 0, 150, 70), opacity=0.9, fill=(0.3, 0.3, 0.9, 0.5)))]),
    "overlay_first_odd": (203, 117, True, [("img_bgra_bgra", 77, 41, dict()), ("img_nv12_bgra", 64, 36, dict(rect=(12, 8, 90, 50), opacity=0.7)),
                                           ("img_bgra_bgra", 203, 117, dict()), ("img_nv12_bgra", 64, 36, dict(rect=(60, 40, 90, 50)))]),
}


@pytest.mark.parametrize("case", list(METAL_CASES))
def test_reference_default_bgra_overlays_inside_wave_ticks(ctx, path, case):
    cw, ch, clear, specs = METAL_CASES[case]
    assert run_tick_case(ctx, cw, ch, clear, specs, expect=None) == WAVE


def test_only_reference_default_bgra_layers_take_the_general_kernel(ctx):
    specs = [("img_bgra_bgra", 96, 54, dict()), ("img_bgra_bgra", 40, 30, dict())]
    assert run_tick_case(ctx, 96, 54, True, specs, expect=None) == "tick_general_bgra"


@pytest.mark.parametrize("seed", range(32))
def test_random_mixed_ticks(ctx, path, seed):
    """Seeded random ticks: 1..8 layers of random kinds with random axis-aligned geometry (placement, crop, flips, borders,
    fill, opacity, up- and downscales), three ticks of different sizes per launch."""
    rng = np.random.default_rng(9000 + seed)
    clear = bool(rng.integers(0, 2))
    kinds = ["img_nv12_bgra", "img_y420p_bgra", "img_bgra_bgra_tx", "img_rgba_bgra_tx"]
    ticks, exps, gds = [], [], []
    for t in range(3):
        cw, ch = int(rng.integers(8, 330)), int(rng.integers(4, 140))
        canvas0 = util.alloc_image("bgra", cw, ch, seed=int(rng.integers(1, 1 << 20)))
        exp = util.copy_image(canvas0)
        if clear:
            assert O.run_kernel("img_clear_bgra", exp) == 0
        layers = []
        for l in range(int(rng.integers(1, 9))):
            k = kinds[int(rng.integers(0, 4))]
            s = k.split("_")[1]
            sw, sh = int(rng.integers(8, 200)) * 2, int(rng.integers(2, 90)) * 2      # rows of >= 16 bytes in every plane
            kw = {}
            if rng.random() < 0.7:
                kw["rect"] = (float(rng.uniform(-0.3, 0.6) * cw), float(rng.uniform(-0.3, 0.6) * ch),
                              float(rng.uniform(0.2, 1.5) * cw), float(rng.uniform(0.2, 1.5) * ch))
            if rng.random() < 0.3:
                kw["border"] = tuple(float(v) for v in rng.uniform(0, 10, 4))
            if rng.random() < 0.3:
                kw["fill"] = tuple(float(v) for v in rng.uniform(0, 1, 4))
            if rng.random() < 0.4:
                kw["tex"] = (float(rng.uniform(0.0, 0.4)), float(rng.uniform(0.0, 0.4)),
                             float(rng.uniform(0.3, 1.0)) * (1 if rng.random() < 0.8 else -1), float(rng.uniform(0.3, 1.0)))
            kw["opacity"] = float(rng.choice([1.0, 1.0, rng.uniform(0, 1)]))
            if seed % 4 == 3 and rng.random() < 0.25:
                kw["rotation"] = float(rng.uniform(-0.8, 0.8))
            u = util.make_uniforms((cw, ch), in_size=(sw, sh), **kw)
            src = util.alloc_image(s, sw, sh, seed=int(rng.integers(1, 1 << 20)))
            csc = int(rng.integers(0, 4))
            assert O.run_kernel(k, exp, src, u, csc=csc, threads=4) == 0
            layers.append((sv.defaultComputeKernelFromString(k), G.to_gpu(ctx, s, sw, sh, src), u, csc))
        gd = G.to_gpu(ctx, "bgra", cw, ch, canvas0)
        ticks.append((gd, clear, layers))
        exps.append(exp)
        gds.append((gd, cw, ch))
    h, name, keep = G.make_batch(ctx, ticks)
    G.run_batch(ctx, h)
    G.destroy_batch(h)
    for i, ((gd, cw, ch), exp) in enumerate(zip(gds, exps)):
        G.assert_same(G.from_gpu(ctx, gd, "bgra", cw, ch), exp, f"seed {seed} tick {i} ({len(ticks[i][2])} layers) via {name}")


@pytest.mark.parametrize("seed", range(24))
def test_random_rgb_only_ticks(ctx, path, seed):
    """Launches of RGB layers only — the instantiation of tick_bgra_wave that fills interior rectangles by LDS-DMA (kernels_wave.hip.cpp,
    CHV_WAVE_DMA): canvases several strips wide and tall, so that strips lie inside a layer (DMA), on its edge and across the picture's edge
    (the register path), layers of BGRA (DMA) and RGBA (byte swap: register path) pictures in one tick, stacks that share their predecessor's
    geometry, up- and downscales between 3:1 and 1:3 (rows per DMA instruction from 1 to 16, the pair form beyond 1.6:1)."""
    rng = np.random.default_rng(9500 + seed)
    clear = bool(rng.integers(0, 2))
    ticks, exps, gds = [], [], []
    for t in range(2):
        cw, ch = int(rng.integers(130, 420)), int(rng.integers(40, 150))
        canvas0 = util.alloc_image("bgra", cw, ch, seed=int(rng.integers(1, 1 << 20)))
        exp = util.copy_image(canvas0)
        if clear:
            assert O.run_kernel("img_clear_bgra", exp) == 0
        layers = []
        u = sw = sh = None
        for l in range(int(rng.integers(1, 7))):
            k = "img_bgra_bgra_tx" if rng.random() < 0.75 else "img_rgba_bgra_tx"
            if u is None or rng.random() < 0.5:
                scale = float(np.exp(rng.uniform(np.log(1 / 3), np.log(3))))
                sw = max(8, int(cw * scale * rng.uniform(0.7, 1.2)) // 4 * 4)
                sh = max(4, int(ch * scale * rng.uniform(0.7, 1.2)) // 2 * 2)
                kw = {"opacity": float(rng.choice([1.0, rng.uniform(0, 1)]))}
                if rng.random() < 0.6:
                    kw["rect"] = (float(rng.uniform(-0.2, 0.3) * cw), float(rng.uniform(-0.2, 0.3) * ch),
                                  float(rng.uniform(0.6, 1.4) * cw), float(rng.uniform(0.6, 1.4) * ch))
                if rng.random() < 0.3:
                    kw["tex"] = (float(rng.uniform(0.0, 0.3)), float(rng.uniform(0.0, 0.3)),
                                 float(rng.uniform(0.5, 1.0)) * (1 if rng.random() < 0.8 else -1), float(rng.uniform(0.5, 1.0)))
                if rng.random() < 0.15:
                    kw["fill"] = tuple(float(v) for v in rng.uniform(0, 1, 4))
                u = util.make_uniforms((cw, ch), in_size=(sw, sh), **kw)
            src = util.alloc_image("bgra", sw, sh, seed=int(rng.integers(1, 1 << 20)))
            assert O.run_kernel(k, exp, src, u, threads=4) == 0
            layers.append((sv.defaultComputeKernelFromString(k), G.to_gpu(ctx, "bgra", sw, sh, src), u, 0))
        gd = G.to_gpu(ctx, "bgra", cw, ch, canvas0)
        ticks.append((gd, clear, layers))
        exps.append(exp)
        gds.append((gd, cw, ch))
    h, name, keep = G.make_batch(ctx, ticks)
    assert name == WAVE, name
    G.run_batch(ctx, h)
    G.destroy_batch(h)
    for i, ((gd, cw, ch), exp) in enumerate(zip(gds, exps)):
        G.assert_same(G.from_gpu(ctx, gd, "bgra", cw, ch), exp, f"seed {seed} tick {i} ({len(ticks[i][2])} layers)")


@pytest.mark.parametrize("kernel", ["tick_bgra_stream", WAVE])
def test_pipeline_full_size(ctx, switch, kernel):
    """The headline tick at full size: 4 x 1080p NV12 -> 720p BGRA canvas, opacities 1/.75/.5/.25 == oracle's clear + 4 kernel calls;
    fused == the sequence of chv_run_kernel launches the reference would issue; replay is idempotent.  Through the streaming kernel (the
    default route of this tick in launches of every size) and through the strip kernel (CHV_STREAM=0; the route of every stack the
    streaming kernel does not take)."""
    if kernel == WAVE:
        switch("CHV_STREAM", "0")
    sw, sh, dw, dh = 1920, 1080, 1280, 720
    srcs = [util.alloc_image("nv12", sw, sh, seed=0x5EED0000 + 48 + i) for i in range(4)]
    us = [util.full_canvas_uniforms((dw, dh), (sw, sh), opacity=o) for o in (1.0, 0.75, 0.5, 0.25)]
    exp = util.alloc_image("bgra", dw, dh)
    assert O.run_kernel("img_clear_bgra", exp, threads=16) == 0
    for s, u in zip(srcs, us):
        assert O.run_kernel("img_nv12_bgra", exp, s, u, threads=16) == 0
    gs = [G.to_gpu(ctx, "nv12", sw, sh, s) for s in srcs]
    gd = G.to_gpu(ctx, "bgra", dw, dh, util.alloc_image("bgra", dw, dh, seed=5))
    layers = [(sv.ComputeKernel.img_nv12_bgra, g, u, 0) for g, u in zip(gs, us)]
    h, name, keep = G.make_batch(ctx, [(gd, True, layers)])
    assert name == kernel
    G.run_batch(ctx, h)
    G.run_batch(ctx, h)
    G.destroy_batch(h)
    G.assert_same(G.from_gpu(ctx, gd, "bgra", dw, dh), exp, "pipeline, fused")
    # the reference's own launch sequence: clear, then one runComputeKernel(blends: true) per layer
    gd2 = G.to_gpu(ctx, "bgra", dw, dh, util.alloc_image("bgra", dw, dh, seed=6))

    def seq(c):
        sv.runComputeKernel(c, images=[], target=gd2, kernel=sv.ComputeKernel.img_clear_bgra)
        for g, u in zip(gs, us):
            sv.runComputeKernel(c, images=[g], target=gd2, kernel=sv.ComputeKernel.img_nv12_bgra, uniforms=u, blends=True)
        return c
    sv.usingContext(ctx, seq)
    G.assert_same(G.from_gpu(ctx, gd2, "bgra", dw, dh), exp, "pipeline, sequential")


def test_mixed_full_size_1080p_canvas(ctx):
    """1080p canvas, four kinds of sources at different scales, in one tick"""
    cw, ch = 1920, 1080
    specs = [("img_nv12_bgra", 1280, 720, dict()),
             ("img_y420p_bgra", 1920, 1080, dict(rect=(960, 0, 960, 540), opacity=0.9)),
             ("img_bgra_bgra_tx", 640, 360, dict(rect=(64, 64, 640, 360), opacity=0.8)),
             ("img_rgba_bgra_tx", 640, 360, dict(rect=(1200, 640, 640, 360), opacity=0.6, border=(8, 8, 8, 8), fill=(0.9, 0.9, 0.9, 0.5)))]
    run_tick_case(ctx, cw, ch, True, specs, seed=77, expect=WAVE)


@pytest.mark.parametrize("target", ["bgra", "nv12"])
def test_composite_more_than_sixteen_layers(ctx, target):
    """chv_composite splits a tick deeper than CHV_MAX_LAYERS into several launches; a mixer composes any number of layers
    (mix.video.swift:116-124)"""
    cw, ch = 160, 64
    src_kinds = ["nv12", "bgra", "y420p", "rgba"] if target == "bgra" else ["nv12", "bgra", "y420p", "rgba"]
    exp = util.alloc_image(target, cw, ch)
    assert O.run_kernel(f"img_clear_{target}", exp) == 0
    layers = []
    for i in range(21):
        s = src_kinds[i % 4]
        k = f"img_{s}_{target}" + ("_tx" if (target == "bgra" and s in ("bgra", "rgba")) else "")
        u = util.make_uniforms((cw, ch), rect=(5 * i, i, 64, 36), opacity=0.95 - 0.03 * i, in_size=(64, 36))
        src = util.alloc_image(s, 64, 36, seed=300 + i)
        assert O.run_kernel(k, exp, src, u) == 0
        layers.append((sv.defaultComputeKernelFromString(k), G.to_gpu(ctx, s, 64, 36, src), u, 0))
    gd = G.to_gpu(ctx, target, cw, ch, util.alloc_image(target, cw, ch, seed=9))
    sv.usingContext(ctx, lambda c: sv.compositeTick(c, gd, layers, True))
    G.assert_same(G.from_gpu(ctx, gd, target, cw, ch), exp, f"21 layers onto {target}")
