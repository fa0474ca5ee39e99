"""tick_yuv_stream (kernels_stream_yuv.hip.cpp) — the reference's own kernels on 4:2:0 canvases (img_nv12_nv12, img_y420p_nv12,
img_y420p_y420p, img_{bgra,rgba}_{nv12,y420p}, kernels.cl.swift:186-255,267-335,469-532) and the integer RGB -> YUV kernels, canvas rows
outermost with every layer's source rows streamed through LDS rings — gives exactly the bytes of the oracle's clear + per-layer kernel
calls: layers of different geometry in one tick, strips and chunks crossed by picture edges, ragged last strips and chunks, reductions and
enlargements up to the rings' limits, every colourspace, batches and lone ticks (descriptors as kernel arguments), random ticks."""
import numpy as np
import pytest

import gpuutil as G
import util
from oracle import oracle as O
from swiftvideo_amd import compute as sv

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def every_eligible_launch(switch):
    """this file is about the streaming kernel: every launch it can take goes through it (the default route takes it only where it measured
    faster: test_default_route)"""
    switch("CHV_YUV_STREAM", "force")


def run_tick(ctx, d, cw, ch, specs, seed=81, csc=0, expect="stream", lone=False, clear=True):
    canvas0 = util.alloc_image(d, cw, ch, seed=seed)
    exp = util.copy_image(canvas0)
    if clear:
        assert O.run_kernel(f"img_clear_{d}", exp) == 0
    layers = []
    for i, (k, sw, sh, kw) in enumerate(specs):
        u = util.make_uniforms((cw, ch), in_size=(sw, sh), **kw)
        s = k.split("_")[1]
        src = util.alloc_image(s, sw, sh, seed=seed + 9 + i)
        assert O.run_kernel(k, exp, src, u, csc=csc, threads=4) == 0
        layers.append((sv.defaultComputeKernelFromString(k), G.to_gpu(ctx, s, sw, sh, src), u, csc))
    gd = G.to_gpu(ctx, d, cw, ch, canvas0)
    h, name, keep = G.make_batch(ctx, [(gd, clear, layers)])
    if expect == "stream":
        assert name == f"tick_yuv_stream<{d}>", name
    elif expect is not None:
        assert name == expect, name
    if lone:
        G.destroy_batch(h)
        sv.usingContext(ctx, lambda c: sv.compositeTick(c, gd, layers, clear))
    else:
        G.run_batch(ctx, h)
        G.destroy_batch(h)
    G.assert_same(G.from_gpu(ctx, gd, d, cw, ch), exp, f"{'lone tick' if lone else 'batch'} via {name}")
    return name


CASES = {
    # name: (target, canvas w, h, [(kernel, src w, h, make_uniforms kwargs)])
    "nv12_copy":         ("nv12", 256, 64, [("img_nv12_nv12", 256, 64, dict())]),
    "y420p_copy":        ("y420p", 256, 64, [("img_y420p_y420p", 256, 64, dict())]),
    "y420p_to_nv12":     ("nv12", 256, 64, [("img_y420p_nv12", 256, 64, dict(opacity=0.7))]),
    "nv12_down_1.5":     ("nv12", 320, 180, [("img_nv12_nv12", 480, 270, dict())]),
    "y420p_down_1.5":    ("y420p", 320, 180, [("img_y420p_y420p", 480, 270, dict(opacity=0.5))]),
    "nv12_up_3x":        ("nv12", 288, 88, [("img_nv12_nv12", 96, 30, dict(opacity=0.5))]),
    "y420p_up_2x":       ("y420p", 384, 112, [("img_y420p_y420p", 192, 56, dict())]),
    "y420p_tall":        ("y420p", 128, 260, [("img_y420p_y420p", 128, 260, dict())]),                       # 16 full groups + one trip
    "nv12_ragged":       ("nv12", 200, 44, [("img_nv12_nv12", 208, 44, dict(opacity=0.9))]),                 # last strip: 8 columns; 2.75 groups
    "y420p_anisotropic": ("y420p", 256, 40, [("img_y420p_y420p", 384, 84, dict())]),                         # 1.5 across, 2.1 down
    "bgra_full":         ("nv12", 192, 64, [("img_bgra_nv12", 192, 64, dict())]),
    "rgba_full_y420p":   ("y420p", 192, 64, [("img_rgba_y420p", 192, 64, dict(opacity=0.8))]),
    "rgba_up":           ("y420p", 192, 64, [("img_rgba_y420p", 100, 36, dict(opacity=0.6))]),
    "rgb_down_vertical": ("nv12", 192, 64, [("img_bgra_nv12", 200, 134, dict())]),                           # 1.04 across, 2.1 down
    "mixer":             ("y420p", 384, 216, [("img_y420p_y420p", 384, 216, dict()),
                                              ("img_bgra_y420p", 128, 72, dict(rect=(16, 16, 128, 72), opacity=0.8)),
                                              ("img_rgba_y420p", 128, 72, dict(rect=(240, 128, 128, 72), opacity=0.6))]),
    "mixer_nv12":        ("nv12", 384, 216, [("img_nv12_nv12", 384, 216, dict()),
                                             ("img_bgra_nv12", 128, 72, dict(rect=(17, 15, 128, 72), opacity=0.8)),
                                             ("img_y420p_nv12", 128, 72, dict(rect=(241, 129, 130, 74), opacity=0.6, border=(4, 4, 4, 4)))]),
    "two_videos":        ("y420p", 320, 180, [("img_y420p_y420p", 480, 270, dict()), ("img_y420p_y420p", 320, 180, dict(opacity=0.5))]),
    "pip":               ("nv12", 320, 180, [("img_nv12_nv12", 320, 180, dict()), ("img_nv12_nv12", 160, 96, dict(rect=(150, 70, 160, 96)))]),
    "four_layers":       ("y420p", 320, 96, [("img_y420p_y420p", 320, 96, dict()), ("img_bgra_y420p", 64, 36, dict(rect=(10, 5, 64, 36), opacity=0.8)),
                                             ("img_bgra_y420p", 160, 64, dict(rect=(120, 20, 160, 64), opacity=0.5)),
                                             ("img_rgba_y420p", 64, 36, dict(rect=(250, 50, 64, 36)))]),
    "three_videos":      ("nv12", 320, 96, [("img_nv12_nv12", 320, 96, dict()), ("img_y420p_nv12", 160, 64, dict(rect=(120, 20, 160, 64), opacity=0.5)),
                                            ("img_nv12_nv12", 96, 48, dict(rect=(10, 40, 96, 48), opacity=0.8))]),
    "off_canvas":        ("nv12", 256, 64, [("img_nv12_nv12", 256, 64, dict()), ("img_bgra_nv12", 64, 36, dict(rect=(-30, -20, 64, 36), opacity=0.6)),
                                            ("img_nv12_nv12", 64, 32, dict(rect=(230, 50, 64, 32), opacity=0.7)), ("img_rgba_nv12", 32, 16, dict(rect=(400, 10, 32, 16)))]),
    "border_no_fill":    ("y420p", 256, 72, [("img_y420p_y420p", 96, 54, dict(rect=(33, 9, 180, 40), border=(5, 3, 7, 2), opacity=0.8)),
                                             ("img_bgra_y420p", 52, 40, dict(rect=(-20, -10, 120, 60), opacity=0.35))]),
    # tex = (x, y, w, h): the picture occupies that part of its rectangle (the border quad is the whole rectangle)
    "cropped":           ("nv12", 192, 64, [("img_nv12_nv12", 96, 64, dict(tex=(0.25, 0.0, 0.5, 1.0))),
                                            ("img_rgba_nv12", 48, 36, dict(rect=(20, 4, 96, 56), tex=(0.2, 0.1, 0.5, 0.7), opacity=0.5))]),
    "opacity_gt_1":      ("y420p", 128, 32, [("img_y420p_y420p", 128, 32, dict(opacity=1.7)), ("img_bgra_y420p", 128, 32, dict(opacity=-0.3))]),
    "odd_rect":          ("y420p", 136, 36, [("img_y420p_y420p", 192, 60, dict(rect=(3, 1, 129, 33))), ("img_rgba_y420p", 68, 36, dict(rect=(31, 7, 67, 25), opacity=0.5))]),
    "int_encoder":       ("nv12", 256, 64, [("img_bgra_nv12_int", 256, 64, dict())]),
    "int_encoder_y420p": ("y420p", 256, 96, [("img_rgba_y420p_int", 256, 96, dict(opacity=0.9))]),
    "int_up":            ("nv12", 288, 88, [("img_bgra_nv12_int", 100, 32, dict(opacity=0.5))]),
    "int_overlays":      ("y420p", 320, 180, [("img_y420p_y420p", 480, 270, dict()),
                                              ("img_bgra_y420p_int", 96, 54, dict(rect=(30, 20, 96, 54), opacity=0.8)),
                                              ("img_rgba_y420p_int", 96, 54, dict(rect=(150, 60, 120, 80), opacity=0.6, border=(4, 4, 4, 4))),
                                              ("img_bgra_y420p", 64, 36, dict(rect=(200, 10, 64, 36)))]),
    "int_edges":         ("y420p", 264, 72, [("img_y420p_y420p", 288, 72, dict()), ("img_bgra_y420p_int", 52, 40, dict(rect=(-20, -10, 120, 60), opacity=0.6)),
                                             ("img_rgba_y420p_int", 64, 36, dict(rect=(131, 33, 101, 31)))]),
}


@pytest.mark.parametrize("lone", [False, True], ids=["batch", "lone"])
@pytest.mark.parametrize("case", list(CASES))
def test_yuv_stream_matches_oracle(ctx, case, lone):
    d, cw, ch, specs = CASES[case]
    run_tick(ctx, d, cw, ch, specs, lone=lone)


@pytest.mark.parametrize("csc", [1, 2, 3])
@pytest.mark.parametrize("case", ["int_encoder", "int_overlays", "int_edges"])
def test_integer_matrix_rows_in_every_colourspace(ctx, case, csc):
    """BT.601 limited is csc 0 (every other case); BT.709 limited, BT.601 full, BT.709 full here"""
    d, cw, ch, specs = CASES[case]
    run_tick(ctx, d, cw, ch, specs, csc=csc)


@pytest.mark.parametrize("case", ["mixer", "four_layers", "y420p_down_1.5", "int_overlays", "nv12_ragged"])
def test_yuv_stream_equals_the_strip_kernel(ctx, switch, case):
    """the same tick with CHV_YUV_STREAM=0: tick_yuv_wave, held to the same oracle bytes"""
    switch("CHV_YUV_STREAM", "0")
    d, cw, ch, specs = CASES[case]
    assert run_tick(ctx, d, cw, ch, specs, expect=None) == f"tick_yuv_wave<{d}>"
    switch("CHV_YUV_STREAM", "force")


def test_default_route(ctx, switch):
    """CHV_YUV_STREAM unset: the streaming kernel takes the launches it measured faster on — all-integer-matrix RGB launches (the encoder side)
    and lone ticks of video layers — and leaves the rest to the strip kernel; both routes give the oracle's bytes"""
    switch("CHV_YUV_STREAM", None)
    enc = [("img_bgra_nv12_int", 256, 64, dict())]
    video = [("img_nv12_nv12", 256, 64, dict())]
    mixer = video + [("img_bgra_nv12", 64, 36, dict(rect=(10, 5, 64, 36), opacity=0.8))]
    assert run_tick(ctx, "nv12", 256, 64, enc, expect=None) == "tick_yuv_stream<nv12>"
    assert run_tick(ctx, "nv12", 256, 64, video, expect=None) == "tick_yuv_wave<nv12>"          # (a batch; the lone tick below streams)
    assert run_tick(ctx, "nv12", 256, 64, mixer, expect=None) == "tick_yuv_wave<nv12>"
    for specs in (enc, video, mixer):
        run_tick(ctx, "nv12", 256, 64, specs, expect=None, lone=True)


def test_yuv_stream_eligibility(ctx):
    """what the streaming kernel declines keeps the strip kernel (or the general one): un-cleared canvases, fill paint, flips, canvases that are
    not a multiple of 8 x 4, reductions beyond the rings, five layers, rings of more than 16 KB per wave, rotation"""
    own = ("img_y420p_y420p", 256, 64, dict())
    wave = "tick_yuv_wave<y420p>"
    assert run_tick(ctx, "y420p", 256, 64, [own], clear=False, expect=None) == wave
    assert run_tick(ctx, "y420p", 256, 64, [own, ("img_bgra_y420p", 64, 36, dict(rect=(10, 5, 64, 36), fill=(0.2, 0.9, 0.1, 0.5)))], expect=None) == wave
    assert run_tick(ctx, "y420p", 256, 64, [("img_y420p_y420p", 256, 64, dict(tex=(1.0, 0.0, -1.0, 1.0)))], expect=None) == wave
    assert run_tick(ctx, "y420p", 260, 64, [("img_y420p_y420p", 256, 64, dict())], expect=None) == wave
    assert run_tick(ctx, "y420p", 256, 66, [("img_y420p_y420p", 256, 64, dict())], expect=None) == wave
    assert run_tick(ctx, "y420p", 128, 64, [("img_y420p_y420p", 256, 64, dict())], expect=None) == wave                # 2 : 1 across
    assert run_tick(ctx, "y420p", 256, 40, [("img_y420p_y420p", 256, 96, dict())], expect=None) == wave                # 2.4 : 1 down
    # between the tested 2.1 and the ring's arithmetic limit 15 / 7: an 8-row step may span 17 source rows at 2.2 — declined
    assert run_tick(ctx, "y420p", 256, 40, [("img_y420p_y420p", 256, 86, dict())], expect=None) == wave                # 2.15 : 1 down
    assert run_tick(ctx, "y420p", 256, 40, [("img_y420p_y420p", 256, 88, dict())], expect=None) == wave                # 2.2 : 1 down
    for sh in (86, 88):                                                                                                # ... also as lone ticks (the default transient route)
        run_tick(ctx, "y420p", 256, 40, [("img_y420p_y420p", 256, sh, dict())], expect=None, lone=True)
    run_tick(ctx, "nv12", 1920, 496, [("img_nv12_nv12", 1920, 1080, dict())], expect=None, lone=True)                    # the advisor's example: 1080 rows into 496
    assert run_tick(ctx, "y420p", 192, 64, [("img_bgra_y420p", 256, 64, dict())], expect=None) == wave                  # RGB texels: 1.33 across
    assert run_tick(ctx, "y420p", 256, 64, [own] * 5, expect=None) == wave
    assert run_tick(ctx, "y420p", 256, 64, [own] * 4, expect=None) == wave
    assert run_tick(ctx, "y420p", 256, 64, [own, ("img_bgra_y420p", 64, 36, dict(rect=(10, 5, 64, 36), rotation=0.3))], expect=None) == wave
    assert run_tick(ctx, "y420p", 256, 64, [("img_y420p_y420p", 200, 64, dict())], expect=None) == wave                # chroma rows of 100 bytes


@pytest.mark.parametrize("d", ["nv12", "y420p"])
def test_batches_of_ticks_of_different_sizes_and_depths(ctx, d):
    """one launch: ticks of different canvas sizes, layer counts and source classes (the launch runs the instantiation for the union)"""
    own = f"img_{d}_{d}"
    specs = [(256, 64, [(own, 256, 64, dict())]),
             (320, 180, [(own, 480, 270, dict()), (f"img_bgra_{d}", 96, 54, dict(rect=(30, 20, 96, 54), opacity=0.8))]),
             (64, 16, [(f"img_rgba_{d}", 64, 16, dict(opacity=0.5))]),
             (384, 216, [(own, 384, 216, dict()), (f"img_bgra_{d}_int", 128, 72, dict(rect=(16, 16, 128, 72), opacity=0.8)),
                         (f"img_rgba_{d}", 128, 72, dict(rect=(240, 128, 128, 72), opacity=0.6)), (f"img_bgra_{d}", 96, 48, dict(rect=(100, 100, 96, 48), opacity=0.4))])]
    ticks, exps, gds = [], [], []
    for t, (cw, ch, ls) in enumerate(specs):
        canvas0 = util.alloc_image(d, cw, ch, seed=300 + t)
        exp = util.copy_image(canvas0)
        assert O.run_kernel(f"img_clear_{d}", exp) == 0
        layers = []
        for i, (k, sw, sh, kw) in enumerate(ls):
            u = util.make_uniforms((cw, ch), in_size=(sw, sh), **kw)
            s = k.split("_")[1]
            src = util.alloc_image(s, sw, sh, seed=400 + 10 * t + i)
            assert O.run_kernel(k, exp, src, u, threads=4) == 0
            layers.append((sv.defaultComputeKernelFromString(k), G.to_gpu(ctx, s, sw, sh, src), u, 0))
        gd = G.to_gpu(ctx, d, cw, ch, canvas0)
        ticks.append((gd, True, layers)); exps.append(exp); gds.append((gd, cw, ch))
    h, name, keep = G.make_batch(ctx, ticks)
    assert name == f"tick_yuv_stream<{d}>", name
    G.run_batch(ctx, h)
    G.destroy_batch(h)
    for t, ((gd, cw, ch), exp) in enumerate(zip(gds, exps)):
        G.assert_same(G.from_gpu(ctx, gd, d, cw, ch), exp, f"tick {t}")


def random_tick(rng, d, integer):
    """an eligible random tick: canvas a multiple of 8 x 4, 1..4 layers of random source kinds and random axis-aligned geometry inside the
    rings' limits (<= 1.6 across for video, native or enlarged for RGB texels, <= 2.1 down), no fill paint, no flips"""
    cw, ch = int(rng.integers(3, 60)) * 8, int(rng.integers(2, 70)) * 4
    kinds = {"nv12": ["img_nv12_nv12", "img_y420p_nv12", "img_bgra_nv12", "img_rgba_nv12"],
             "y420p": ["img_y420p_y420p", "img_bgra_y420p", "img_rgba_y420p"]}[d]
    if integer:
        kinds = [k + "_int" if k.split("_")[1] in ("bgra", "rgba") else k for k in kinds] + [f"img_bgra_{d}_int"]
    specs = []
    lds_bytes = 0
    for l in range(int(rng.integers(1, 5))):
        k = kinds[int(rng.integers(0, len(kinds)))]
        rgb = k.split("_")[1] in ("bgra", "rgba")
        # (a wave's rings: 5.4 KB per video layer, 2.5 KB per RGB layer, 1.7 KB of row tables, 16 KB per wave)
        lds_bytes += 2560 if rgb else 5376
        if lds_bytes + 4 * 416 > 16384:
            break
        kw = {}
        full = l == 0 and rng.random() < 0.6
        rx, ry_ = 0.0, 0.0
        if full:
            rw, rh = float(cw), float(ch)
        else:
            rw, rh = float(rng.uniform(0.15, 1.3) * cw), float(rng.uniform(0.15, 1.3) * ch)
            rx, ry_ = float(rng.uniform(-0.3, 0.8) * cw), float(rng.uniform(-0.3, 0.8) * ch)
        tw, th = 1.0, 1.0
        if rng.random() < 0.3:
            tw, th = float(rng.uniform(0.4, 1.0)), float(rng.uniform(0.4, 1.0))
            kw["tex"] = (float(rng.uniform(0.0, 1.0 - tw)), float(rng.uniform(0.0, 1.0 - th)), tw, th)
        # source size: the picture covers tex.w x tex.h of its rectangle; source rows are a multiple of 16 bytes in every plane
        unit, lim = (4, 1.12) if rgb else (32, 1.6)
        kx = float(rng.uniform(0.3, lim - 0.02)) if rng.random() < 0.7 else 1.0
        ky = float(rng.uniform(0.3, 2.05)) if rng.random() < 0.7 else 1.0
        sw = max(unit, int(kx * rw * tw) // unit * unit)
        sh = max(4, int(ky * rh * th) // 2 * 2)
        if sw / (rw * tw) > lim or sh / (rh * th) > 2.1:
            # (tiny rectangles: the smallest source is still too large for them) draw it larger
            rw, rh = max(rw, sw / tw / lim * 1.01), max(rh, sh / th / 2.1 * 1.01)
            full = False
        if not full:
            kw["rect"] = (rx, ry_, rw, rh)
        if rng.random() < 0.3:
            kw["border"] = tuple(float(v) for v in rng.uniform(0, 10, 4))
        kw["opacity"] = float(rng.choice([1.0, 1.0, rng.uniform(0, 1)]))
        specs.append((k, sw, sh, kw, int(rng.integers(0, 4)) if integer else 0))
    return cw, ch, specs


def build_random(ctx, rng, d, integer):
    cw, ch, specs = random_tick(rng, d, integer)
    canvas0 = util.alloc_image(d, cw, ch, seed=int(rng.integers(1, 1 << 20)))
    exp = util.copy_image(canvas0)
    assert O.run_kernel(f"img_clear_{d}", exp) == 0
    layers = []
    for k, sw, sh, kw, csc in specs:
        u = util.make_uniforms((cw, ch), in_size=(sw, sh), **kw)
        s = k.split("_")[1]
        src = util.alloc_image(s, sw, sh, seed=int(rng.integers(1, 1 << 20)))
        assert O.run_kernel(k, exp, src, u, csc=csc, threads=4) == 0
        layers.append((sv.defaultComputeKernelFromString(k), G.to_gpu(ctx, s, sw, sh, src), u, csc))
    gd = G.to_gpu(ctx, d, cw, ch, canvas0)
    return gd, cw, ch, layers, exp, specs


@pytest.mark.parametrize("seed", list(range(40)) + [f"int{i}" for i in range(24)])
def test_random_yuv_stream_ticks(ctx, seed):
    """Seeded random eligible ticks, three per launch.  Seeds `int<n>`: the RGB layers through the integer-matrix kernels, a random colourspace each."""
    integer = isinstance(seed, str)
    seed = int(seed[3:]) + 100 if integer else seed
    rng = np.random.default_rng(21000 + seed)
    d = "nv12" if seed % 2 == 0 else "y420p"
    built = [build_random(ctx, rng, d, integer) for _ in range(3)]
    h, name, keep = G.make_batch(ctx, [(b[0], True, b[3]) for b in built])
    assert name == f"tick_yuv_stream<{d}>", (name, [b[5] for b in built])
    G.run_batch(ctx, h)
    G.destroy_batch(h)
    for t, (gd, cw, ch, layers, exp, specs) in enumerate(built):
        G.assert_same(G.from_gpu(ctx, gd, d, cw, ch), exp, f"seed {seed} tick {t}: {cw}x{ch} {specs}")


@pytest.mark.parametrize("seed", list(range(24)))
def test_random_lone_yuv_stream_ticks(ctx, seed):
    """the same kind of random ticks, one at a time through chv_composite: descriptors as kernel arguments (tick_yuv_stream_one)"""
    rng = np.random.default_rng(23000 + seed)
    d = "nv12" if seed % 2 == 0 else "y420p"
    gd, cw, ch, layers, exp, specs = build_random(ctx, rng, d, seed % 3 == 2)
    sv.usingContext(ctx, lambda c: sv.compositeTick(c, gd, layers, True))
    G.assert_same(G.from_gpu(ctx, gd, d, cw, ch), exp, f"seed {seed}: {cw}x{ch} {specs}")


@pytest.mark.parametrize("d", ["nv12", "y420p"])
def test_mixer_tick_full_size(ctx, d):
    """the reference-default mixer tick at 1080p: a full-canvas video + two 640 x 360 overlays (bench workload mixer_<d>), as a batch of two
    ticks and as a lone tick"""
    own = f"img_{d}_{d}"
    specs = [(own, 1920, 1080, dict()), (f"img_bgra_{d}", 640, 360, dict(rect=(64, 64, 640, 360), opacity=0.8)),
             (f"img_bgra_{d}", 640, 360, dict(rect=(1200, 640, 640, 360), opacity=0.6))]
    run_tick(ctx, d, 1920, 1080, specs)
    run_tick(ctx, d, 1920, 1080, specs, lone=True, seed=97)


def test_encoder_frame_full_size(ctx):
    """1080p BGRA -> NV12 through the integer matrix (bench workload encode_nv12) and 720p RGBA -> y420p"""
    run_tick(ctx, "nv12", 1920, 1080, [("img_bgra_nv12_int", 1920, 1080, dict())])
    run_tick(ctx, "y420p", 1280, 720, [("img_rgba_y420p_int", 1280, 720, dict())], lone=True)
