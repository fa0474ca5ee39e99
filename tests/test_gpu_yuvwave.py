"""tick_yuv_wave — the reference's own kernels (img_nv12_nv12, img_y420p_nv12, img_y420p_y420p, img_{bgra,rgba}_{nv12,y420p},
kernels.cl.swift:47-532) on 4:2:0 canvases, one wave per canvas strip — gives exactly the bytes of the oracle's clear +
per-layer kernel calls (= the general quad kernel's): strips inside a picture (branch-free path), strips on picture / border
edges and fill paint (per-pixel path), chroma ownership of the even/even pixel, un-cleared canvases, deep ticks."""
import numpy as np
import pytest

import gpuutil as G
import util
from oracle import oracle as O
from swiftvideo_amd import compute as sv

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def strip_kernel_only(switch):
    """this file is about the strip kernel: cleared ticks the streaming kernel would take (tests/test_gpu_yuvstream.py) stay here"""
    switch("CHV_YUV_STREAM", "0")


@pytest.fixture(params=["8", "16"])
def rows(request, switch):
    """both strip heights of tick_yuv_wave (the host picks per launch: 16 rows for launches of >= 8192 strips; CHV_WAVE_ROWS
    forces it for any launch whose 16-row rectangles fit the LDS)"""
    switch("CHV_WAVE_ROWS", request.param)
    return request.param


def run_yuv_tick(ctx, d, cw, ch, clear, specs, seed=81, expect="wave", csc=0):
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
    if expect == "wave":
        assert name == f"tick_yuv_wave<{d}>", name
    G.run_batch(ctx, h)
    G.destroy_batch(h)
    G.assert_same(G.from_gpu(ctx, gd, d, cw, ch), exp, f"via {name}")
    return name


CASES = {
    # name: (target, canvas w, h, clear_first, [(kernel, src w, h, make_uniforms kwargs)])
    "nv12_copy":        ("nv12", 256, 64, True, [("img_nv12_nv12", 256, 64, dict())]),
    "y420p_copy":       ("y420p", 256, 64, True, [("img_y420p_y420p", 256, 64, dict())]),
    "y420p_to_nv12":    ("nv12", 256, 64, True, [("img_y420p_nv12", 256, 64, dict(opacity=0.7))]),
    "nv12_down_1.5":    ("nv12", 320, 180, True, [("img_nv12_nv12", 480, 270, dict())]),
    "y420p_down_1.5":   ("y420p", 320, 180, True, [("img_y420p_y420p", 480, 270, dict(opacity=0.5))]),
    "nv12_up_3x":       ("nv12", 300, 90, True, [("img_nv12_nv12", 100, 30, dict(opacity=0.5))]),
    "bgra_full":        ("nv12", 192, 64, True, [("img_bgra_nv12", 192, 64, dict())]),
    "rgba_full_y420p":  ("y420p", 192, 64, True, [("img_rgba_y420p", 300, 100, dict(opacity=0.8, fill=(0.3, 0.8, 0.1, 0.6)))]),
    "mixer":            ("y420p", 384, 216, True, [("img_y420p_y420p", 384, 216, dict()),
                                                   ("img_bgra_y420p", 128, 72, dict(rect=(16, 16, 128, 72), opacity=0.8)),
                                                   ("img_rgba_y420p", 128, 72, dict(rect=(240, 128, 128, 72), opacity=0.6, fill=(0.2, 0.9, 0.1, 0.5)))]),
    "mixer_nv12":       ("nv12", 384, 216, True, [("img_nv12_nv12", 384, 216, dict()),
                                                  ("img_bgra_nv12", 128, 72, dict(rect=(17, 15, 128, 72), opacity=0.8)),
                                                  ("img_y420p_nv12", 128, 72, dict(rect=(241, 129, 130, 74), opacity=0.6, border=(4, 4, 4, 4), fill=(0.9, 0.2, 0.1, 0.8)))]),
    "rect_border_fill": ("nv12", 260, 70, True, [("img_y420p_nv12", 96, 54, dict(rect=(33, 9, 180, 40), border=(5, 3, 7, 2), fill=(0.9, 0.2, 0.1, 0.6), opacity=0.8)),
                                                 ("img_bgra_nv12", 50, 40, dict(rect=(-20, -10, 120, 60), fill=(0.1, 0.5, 0.9, 1.0), opacity=0.35)),
                                                 ("img_nv12_nv12", 64, 64, dict(rect=(150, 5, 90, 60), tex=(0.25, 0.0, 0.5, 1.0), fill=(1, 1, 0, 1)))]),
    "noclear":          ("y420p", 130, 38, False, [("img_y420p_y420p", 50, 20, dict(rect=(10, 5, 80, 30), opacity=0.5)),
                                                   ("img_bgra_y420p", 50, 20, dict(rect=(60, 2, 60, 36)))]),
    "noclear_full":     ("nv12", 192, 64, False, [("img_nv12_nv12", 192, 64, dict(opacity=0.4)), ("img_rgba_nv12", 192, 64, dict(opacity=0.5))]),
    "flips":            ("nv12", 192, 40, True, [("img_nv12_nv12", 96, 54, dict(tex=(1.0, 0.0, -1.0, 1.0))),
                                                 ("img_rgba_nv12", 96, 54, dict(tex=(0.2, 1.0, 0.5, -0.7), opacity=0.5))]),
    "opacity_gt_1":     ("y420p", 128, 32, True, [("img_y420p_y420p", 128, 32, dict(opacity=1.7)), ("img_bgra_y420p", 128, 32, dict(opacity=-0.3))]),
    "twelve_layers":    ("nv12", 128, 32, True, [("img_bgra_nv12" if i % 3 else "img_nv12_nv12", 64, 16, dict(rect=(4 * i, i, 64, 16), opacity=1.0 - 0.05 * i)) for i in range(12)]),
    "odd_strip_edges":  ("y420p", 130, 22, True, [("img_y420p_y420p", 200, 60, dict()), ("img_rgba_y420p", 66, 34, dict(opacity=0.5))]),
    # LF_SAME_GEOM: a layer whose matrices, plane sizes and bounding box equal its predecessor's keeps that layer's geometry (only
    # opacity, fill colour and the plane pointers differ); interleaved with layers of other geometry / other source classes
    "same_geom":        ("y420p", 320, 180, True, [("img_y420p_y420p", 480, 270, dict()), ("img_y420p_y420p", 480, 270, dict(opacity=0.5)),
                                                   ("img_bgra_y420p", 96, 54, dict(rect=(30, 20, 96, 54), opacity=0.8)),
                                                   ("img_bgra_y420p", 96, 54, dict(rect=(30, 20, 96, 54), opacity=0.4, fill=(0.2, 0.9, 0.1, 0.5))),
                                                   ("img_y420p_y420p", 480, 270, dict(opacity=0.25)),
                                                   ("img_rgba_y420p", 96, 54, dict(rect=(30, 20, 96, 54), opacity=0.4))]),
    "same_geom_nv12":   ("nv12", 256, 64, False, [("img_nv12_nv12", 128, 32, dict(rect=(-20, -6, 200, 50), opacity=0.6)),
                                                  ("img_nv12_nv12", 128, 32, dict(rect=(-20, -6, 200, 50), opacity=0.3)),
                                                  ("img_y420p_nv12", 128, 32, dict(rect=(-20, -6, 200, 50), opacity=0.5)),
                                                  ("img_y420p_nv12", 128, 32, dict(rect=(-20, -6, 200, 50)))]),
    # the integer-matrix RGB kind inside the strip kernel: per-pixel path, in z order with staged layers
    "int_overlays":     ("y420p", 320, 180, True, [("img_y420p_y420p", 480, 270, dict()),
                                                   ("img_bgra_y420p_int", 96, 54, dict(rect=(30, 20, 96, 54), opacity=0.8)),
                                                   ("img_rgba_y420p_int", 96, 54, dict(rect=(150, 60, 120, 80), opacity=0.6, border=(4, 4, 4, 4), fill=(0.2, 0.9, 0.1, 0.5))),
                                                   ("img_bgra_y420p", 64, 36, dict(rect=(200, 10, 64, 36)))]),
    "int_encoder":      ("nv12", 256, 64, True, [("img_bgra_nv12_int", 256, 64, dict())]),
    # its branch-free row loop: whole strips inside the picture with and without fill paint, scaled, flipped, on a canvas that is not
    # cleared (source alpha blends with what is there), strips crossed by the picture's edge (no fill: masked rows; with fill: per pixel)
    "int_full_fill":    ("y420p", 192, 64, True, [("img_rgba_y420p_int", 192, 64, dict(opacity=0.8, fill=(0.3, 0.8, 0.1, 0.6)))]),
    "int_scaled":       ("y420p", 192, 64, True, [("img_rgba_y420p_int", 300, 100, dict(opacity=0.9))]),
    "int_up":           ("nv12", 300, 90, False, [("img_bgra_nv12_int", 100, 30, dict(opacity=0.5))]),
    "int_noclear":      ("nv12", 192, 64, False, [("img_nv12_nv12", 192, 64, dict(opacity=0.4)), ("img_rgba_nv12_int", 192, 64, dict(opacity=0.5)),
                                                  ("img_bgra_nv12_int", 192, 64, dict(opacity=0.7, fill=(0.9, 0.1, 0.4, 0.3)))]),
    "int_flips":        ("nv12", 192, 40, True, [("img_rgba_nv12_int", 96, 54, dict(tex=(0.2, 1.0, 0.5, -0.7), opacity=0.5)),
                                                 ("img_bgra_nv12_int", 96, 54, dict(tex=(1.0, 0.0, -1.0, 1.0), opacity=0.5))]),
    "int_edges":        ("y420p", 260, 70, True, [("img_y420p_y420p", 260, 70, dict()), ("img_bgra_y420p_int", 50, 40, dict(rect=(-20, -10, 120, 60), opacity=0.6)),
                                                  ("img_rgba_y420p_int", 64, 36, dict(rect=(131, 33, 101, 31)))]),
    "int_edges_fill":   ("nv12", 260, 70, True, [("img_bgra_nv12_int", 50, 40, dict(rect=(-20, -10, 120, 60), fill=(0.1, 0.5, 0.9, 1.0), opacity=0.35)),
                                                 ("img_rgba_nv12_int", 96, 54, dict(rect=(33, 9, 180, 40), border=(5, 3, 7, 2), fill=(0.9, 0.2, 0.1, 0.6), opacity=0.8))]),
    "int_opacity_gt_1": ("y420p", 128, 32, True, [("img_bgra_y420p_int", 128, 32, dict(opacity=1.7)), ("img_rgba_y420p_int", 128, 32, dict(opacity=-0.3, fill=(0.5, 0.5, 0.5, 1.0)))]),
    "down_2.5":         ("nv12", 130, 50, True, [("img_nv12_nv12", 326, 124, dict()), ("img_bgra_nv12", 326, 124, dict(opacity=0.5))]),
    # rectangles staged as every row's own pair of tap rows (WGeom::pair, 8-row strips): 3:1 video layers, the same flipped and across the canvas
    # edges, 5:1 (chroma pairs as well), vertical-only reductions, an RGB thumbnail source
    "down_3x_nv12":     ("nv12", 128, 72, True, [("img_nv12_nv12", 384, 216, dict()), ("img_nv12_nv12", 384, 216, dict(rect=(-12, -6, 100, 60), opacity=0.6))]),
    "down_3x_y420p":    ("y420p", 128, 72, True, [("img_y420p_y420p", 384, 216, dict()), ("img_y420p_y420p", 300, 230, dict(tex=(1.0, 1.0, -1.0, -1.0), opacity=0.5))]),
    "down_5x":          ("nv12", 128, 72, False, [("img_y420p_nv12", 640, 360, dict(opacity=0.7)), ("img_nv12_nv12", 640, 400, dict(rect=(20, 10, 90, 50)))]),
    "down_vertical":    ("y420p", 128, 72, True, [("img_y420p_y420p", 128, 216, dict()), ("img_bgra_y420p", 48, 160, dict(rect=(70, 6, 48, 56), opacity=0.8)),
                                                    ("img_rgba_y420p", 60, 190, dict(rect=(4, 8, 60, 60), opacity=0.5))]),
}


@pytest.mark.parametrize("case", list(CASES))
def test_yuv_wave_matches_oracle(ctx, rows, case):
    d, cw, ch, clear, specs = CASES[case]
    # down_2.5: the 4-byte texel rectangles of four waves exceed the LDS budget -> general quad kernel
    run_yuv_tick(ctx, d, cw, ch, clear, specs, expect=None if case == "down_2.5" else "wave")


@pytest.mark.parametrize("csc", [1, 2, 3])
@pytest.mark.parametrize("case", ["int_full_fill", "int_noclear", "int_edges"])
def test_integer_matrix_rows_in_every_colourspace(ctx, rows, case, csc):
    """BT.601 limited is csc 0 (every other case); BT.709 limited, BT.601 full, BT.709 full here"""
    d, cw, ch, clear, specs = CASES[case]
    run_yuv_tick(ctx, d, cw, ch, clear, specs, csc=csc)


@pytest.mark.parametrize("case", ["mixer", "rect_border_fill", "noclear", "flips", "same_geom", "int_noclear", "int_edges", "int_scaled"])
def test_yuv_wave_equals_general_kernel(ctx, switch, case):
    """the same tick through CHV_FORCE_GENERAL=1: both device paths are held to the same oracle bytes"""
    switch("CHV_FORCE_GENERAL", "1")
    d, cw, ch, clear, specs = CASES[case]
    assert run_yuv_tick(ctx, d, cw, ch, clear, specs, expect=None) == f"tick_general_yuv<{d}>"


def test_yuv_wave_fallbacks(ctx):
    """odd canvas sizes (gid/2 leaves the chroma plane), rotated layers, sources too narrow to stage -> general quad kernel"""
    assert run_yuv_tick(ctx, "nv12", 33, 17, True, [("img_bgra_nv12", 64, 36, dict())], expect=None) == "tick_general_yuv<nv12>"
    assert run_yuv_tick(ctx, "y420p", 64, 36, True, [("img_y420p_y420p", 64, 36, dict(rect=(8, 4, 40, 24), rotation=0.3))], expect=None) == "tick_general_yuv<y420p>"
    assert run_yuv_tick(ctx, "y420p", 64, 36, True, [("img_y420p_y420p", 24, 12, dict())], expect=None) == "tick_general_yuv<y420p>"


@pytest.mark.parametrize("d", ["nv12", "y420p"])
def test_rotated_overlay_stays_in_the_wave_kernel(ctx, rows, d):
    """a rotated overlay among axis-aligned layers is applied per pixel inside tick_yuv_wave (z order kept)"""
    own = f"img_{d}_{d}"
    specs = [(own, 384, 216, dict()),
             (f"img_bgra_{d}", 128, 72, dict(rect=(40, 30, 128, 72), rotation=0.35, opacity=0.8, border=(4, 4, 4, 4), fill=(0.2, 0.9, 0.1, 0.5))),
             (f"img_rgba_{d}", 128, 72, dict(rect=(240, 128, 128, 72), opacity=0.6)),
             (own, 96, 54, dict(rect=(150, 20, 120, 70), rotation=-0.5, opacity=0.7))]
    assert run_yuv_tick(ctx, d, 384, 216, True, specs, expect=None) == f"tick_yuv_wave<{d}>"


@pytest.mark.parametrize("seed", list(range(32)) + [f"int{i}" for i in range(16)])
def test_random_yuv_ticks(ctx, rows, seed):
    """Seeded random ticks on 4:2:0 canvases: 1..8 layers of random source kinds with random axis-aligned geometry, three ticks
    of different (even) sizes per launch.  Seeds `int<n>`: the RGB layers through the integer-matrix kernels, a random colourspace each."""
    integer = isinstance(seed, str)
    seed = int(seed[3:]) + 100 if integer else seed
    rng = np.random.default_rng(11000 + seed)
    d = "nv12" if seed % 2 == 0 else "y420p"
    clear = bool(rng.integers(0, 2))
    kinds = {"nv12": ["img_nv12_nv12", "img_y420p_nv12", "img_bgra_nv12", "img_rgba_nv12"],
             "y420p": ["img_y420p_y420p", "img_bgra_y420p", "img_rgba_y420p"]}[d]
    if integer:
        kinds = [k + "_int" if k.split("_")[1] in ("bgra", "rgba") else k for k in kinds] + [f"img_bgra_{d}_int"]
    ticks, exps, gds = [], [], []
    for t in range(3):
        cw, ch = int(rng.integers(4, 165)) * 2, int(rng.integers(2, 70)) * 2
        canvas0 = util.alloc_image(d, cw, ch, seed=int(rng.integers(1, 1 << 20)))
        exp = util.copy_image(canvas0)
        if clear:
            assert O.run_kernel(f"img_clear_{d}", exp) == 0
        layers = []
        for l in range(int(rng.integers(1, 9))):
            k = kinds[int(rng.integers(0, len(kinds)))]
            s = k.split("_")[1]
            sw, sh = int(rng.integers(16, 200)) * 2, int(rng.integers(2, 90)) * 2      # rows of >= 16 bytes in every plane
            kw = {}
            if rng.random() < 0.6:
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
            csc = int(rng.integers(0, 4)) if integer else 0
            assert O.run_kernel(k, exp, src, u, csc=csc, threads=4) == 0
            layers.append((sv.defaultComputeKernelFromString(k), G.to_gpu(ctx, s, sw, sh, src), u, csc))
        gd = G.to_gpu(ctx, d, cw, ch, canvas0)
        ticks.append((gd, clear, layers))
        exps.append(exp)
        gds.append((gd, cw, ch))
    h, name, keep = G.make_batch(ctx, ticks)
    assert name in (f"tick_yuv_wave<{d}>", f"tick_general_yuv<{d}>"), name     # (strong downscales of 4-byte texels exceed the LDS budget)
    G.run_batch(ctx, h)
    G.destroy_batch(h)
    for i, ((gd, cw, ch), exp) in enumerate(zip(gds, exps)):
        G.assert_same(G.from_gpu(ctx, gd, d, cw, ch), exp, f"seed {seed} tick {i} ({len(ticks[i][2])} layers) via {name}")


@pytest.mark.parametrize("d", ["y420p", "nv12"])
def test_reference_default_mixer_canvas_full_size(ctx, rows, d):
    """the bench's mixer workload at full size: 1080p 4:2:0 canvas <- full-canvas 1080p layer + two 640x360 BGRA overlays"""
    cw, ch = 1920, 1080
    specs = [(f"img_{d}_{d}", 1920, 1080, dict()),
             (f"img_bgra_{d}", 640, 360, dict(rect=(64, 64, 640, 360), opacity=0.8)),
             (f"img_bgra_{d}", 640, 360, dict(rect=(1200, 640, 640, 360), opacity=0.6))]
    run_yuv_tick(ctx, d, cw, ch, True, specs, seed=0x5EED0000 + 64)


@pytest.mark.parametrize("cw,pitch", [(130, 130), (132, 132), (128, 134)])
def test_wrapped_canvas_with_tight_pitch(ctx, rows, cw, pitch):
    """A foreign NV12 canvas (one allocation, luma + chroma adjacent, pitch = width or any other value, no alignment promised):
    never a byte outside the payload of a row is written."""
    import ctypes as C
    from swiftvideo_amd import chipvideo as cv
    lib = cv.load()
    ch, sw, sh = 38, 96, 54
    canvas0 = util.alloc_image("nv12", cw, ch, seed=401)
    frame = np.full(pitch * (ch + ch // 2) + 8, 0xA5, dtype=np.uint8)
    frame[: pitch * ch].reshape(ch, pitch)[:, :cw] = canvas0[0]
    frame[pitch * ch: pitch * (ch + ch // 2)].reshape(ch // 2, pitch)[:, :cw] = canvas0[1].reshape(ch // 2, cw)
    owner = C.c_void_p()
    cv.check(lib.chv_buffer_alloc(ctx.handle, frame.size, C.byref(owner)))
    cv.check(lib.chv_upload(ctx.handle, owner, 0, frame.size, frame.ctypes.data, frame.size, frame.size, 1, 0))
    devptr = C.c_void_p()
    cv.check(lib.chv_buffer_info(owner, C.byref(devptr), None))
    wrapped = C.c_void_p()
    cv.check(lib.chv_buffer_wrap(ctx.handle, devptr, frame.size, C.byref(wrapped)))
    img = cv.Image()
    img.format, img.width, img.height, img.n_planes = cv.FMT_NV12, cw, ch, 2
    img.planes[0] = cv.Plane(wrapped.value, 0, cw, ch, pitch, 1)
    img.planes[1] = cv.Plane(wrapped.value, pitch * ch, cw // 2, ch // 2, pitch, 2)
    src = util.alloc_image("nv12", sw, sh, seed=402)
    gsrc = G.to_gpu(ctx, "nv12", sw, sh, src)
    u = util.make_uniforms((cw, ch), in_size=(sw, sh), opacity=0.7)
    sdesc = sv._image_desc(gsrc)
    arr = (cv.Image * 1)(sdesc)
    cv.check(lib.chv_pass_begin(ctx.handle))
    cv.check(lib.chv_run_kernel(ctx.handle, cv.K_IMG_NV12_NV12, C.byref(img), arr, 1, u.ctypes.data, 236, 1, None))
    cv.check(lib.chv_pass_end(ctx.handle, 1))
    exp = util.copy_image(canvas0)
    assert O.run_kernel("img_nv12_nv12", exp, src, u) == 0
    back = np.zeros_like(frame)
    cv.check(lib.chv_download(ctx.handle, back.ctypes.data, back.size, owner, 0, back.size, back.size, 1))
    want = frame.copy()
    want[: pitch * ch].reshape(ch, pitch)[:, :cw] = exp[0]
    want[pitch * ch: pitch * (ch + ch // 2)].reshape(ch // 2, pitch)[:, :cw] = exp[1].reshape(ch // 2, cw)
    assert np.array_equal(back, want), f"first difference at byte {int(np.argwhere(back != want)[0][0])}"
    cv.check(lib.chv_buffer_free(wrapped))
    cv.check(lib.chv_buffer_free(owner))
