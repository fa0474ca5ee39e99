"""Vectors whose expected bytes are derived in the test itself, from the kernels' published formulas and plain integer /
float32 arithmetic — NOT produced by oracle/ref_kernels.c.  They pin the oracle (CPU leg) and the HIP path (`-m gpu` leg)
independently of each other: constant pictures, pure primaries, exact 2x2 box averages, clears, opacity ladders.

What each derivation rests on (reference file:line):
  * clears: Y = 0.0, chroma = 0.5, BGRA = (0,0,0,1)                         kernels.cl.swift:38-46,174-185,257-265
  * UNORM8 store: convert_uchar_sat_rte(f * 255)                            OpenCL 1.2 section 8.3.1.1
  * rgb2yuv rows (0.299, 0.587, 0.113 | -0.169, -0.331, 0.5, +0.5 | 0.5, -0.419, -0.081, +0.5)   kernels.cl.swift:96-99
  * out_uv = gid / size (not pixel centred) => a same-size full-canvas layer samples at gid - 0.5: a 2x2 box average with
    the left / upper neighbour, clamped at the edge                         kernels.cl.swift:70-72 + OpenCL 1.2 section 8.2
  * integer YUV -> RGB: (cy (Y - yoff) + 32768 + c e) >> 16, clipped        DESIGN.md section 4.2
SURVEY section 0.5 records what the reference's own compiled OpenCL kernel produced for an opaque pure-blue BGRA layer on
a cleared NV12 canvas: Y = 29, U = 255, V = 107 — the first vector below."""
import numpy as np
import pytest

import util
from oracle import oracle as O

f32 = np.float32


def st8(v):
    """convert_uchar_sat_rte(f * 255): float32 product, round half to even, saturate"""
    return int(np.clip(np.rint(f32(v) * f32(255)), 0, 255))


def rgb2yuv_codes(r, g, b):
    """(r, g, b) in [0,1] -> stored Y, U, V codes: dot((r,g,b,1), row) summed left to right in float32"""
    rows = [(0.299, 0.587, 0.113, 0.0), (-0.169, -0.331, 0.5, 0.5), (0.5, -0.419, -0.081, 0.5)]
    out = []
    for m in rows:
        acc = f32(r) * f32(m[0]) + f32(g) * f32(m[1])
        acc = f32(acc) + f32(b) * f32(m[2])
        acc = f32(acc) + f32(1.0) * f32(m[3])
        out.append(st8(acc))
    return tuple(out)


CSC = {0: (16, 76309, 104597, 25675, 53279, 132201), 1: (16, 76309, 117489, 13975, 34925, 138438),
       2: (0, 65536, 91881, 22553, 46802, 116130), 3: (0, 65536, 103206, 12276, 30679, 121609)}


def yuv2bgr(csc, y, u, v):
    yoff, cy, crv, cgu, cgv, cbu = CSC[csc]
    c = cy * (y - yoff) + 32768
    d, e = u - 128, v - 128
    clip = lambda t: max(0, min(255, t >> 16))     # noqa: E731  (>> on a negative Python int floors, like an arithmetic shift)
    return clip(c + cbu * d), clip(c - cgu * d - cgv * e), clip(c + crv * e)


PRIMARIES = {"blue": (0, 0, 1), "red": (1, 0, 0), "green": (0, 1, 0), "white": (1, 1, 1), "black": (0, 0, 0), "grey": (0.5, 0.5, 0.5)}


# ---------------------------------------------------------------------------------------------------------------
# runners: the same (kernel, canvas, layers) job through the oracle or through the HIP path
# ---------------------------------------------------------------------------------------------------------------
def run_oracle(target_fmt, cw, ch, layers):
    exp = util.alloc_image(target_fmt, cw, ch)
    assert O.run_kernel(f"img_clear_{target_fmt}", exp) == 0
    for k, src, u, csc in layers:
        assert O.run_kernel(k, exp, src, u, csc=csc) == 0
    return exp


def run_hip(ctx, target_fmt, cw, ch, layers):
    import gpuutil as G
    from swiftvideo_amd import compute as sv
    gd = G.to_gpu(ctx, target_fmt, cw, ch, util.alloc_image(target_fmt, cw, ch, seed=1234))
    gl = []
    for k, src, u, csc in layers:
        s = k.split("_")[1]
        h, w = src[0].shape[0], src[0].shape[1]
        gl.append((sv.defaultComputeKernelFromString(k), G.to_gpu(ctx, s, w, h, src), u, csc))
    sv.usingContext(ctx, lambda c: sv.compositeTick(c, gd, gl, True))
    return G.from_gpu(ctx, gd, target_fmt, cw, ch)


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def run(request):
    if request.param == "oracle":
        return run_oracle
    ctx = request.getfixturevalue("ctx")
    return lambda *a: run_hip(ctx, *a)


def const_image(fmt, w, h, values):
    img = util.alloc_image(fmt, w, h)
    if fmt in ("bgra", "rgba"):
        img[0][...] = np.array(values, dtype=np.uint8)
    elif fmt == "nv12":
        img[0][...] = values[0]; img[1][..., 0] = values[1]; img[1][..., 1] = values[2]
    else:
        img[0][...] = values[0]; img[1][...] = values[1]; img[2][...] = values[2]
    return img


# ---------------------------------------------------------------------------------------------------------------
def test_clears(run):
    nv = run("nv12", 16, 8, [])
    assert np.all(nv[0] == 0) and np.all(nv[1] == 128)          # 0.5 * 255 = 127.5 -> ties to even -> 128
    yp = run("y420p", 16, 8, [])
    assert np.all(yp[0] == 0) and np.all(yp[1] == 128) and np.all(yp[2] == 128)
    bg = run("bgra", 16, 8, [])
    assert np.all(bg[0][..., :3] == 0) and np.all(bg[0][..., 3] == 255)


@pytest.mark.parametrize("name", list(PRIMARIES))
@pytest.mark.parametrize("target", ["nv12", "y420p"])
def test_opaque_primaries_onto_yuv_canvases(run, name, target):
    """an opaque constant BGRA / RGBA picture over the whole canvas: every pixel becomes rgb2yuv(colour)"""
    r, g, b = PRIMARIES[name]
    R, G_, B = (int(round(c * 255)) for c in (r, g, b))
    want = rgb2yuv_codes(f32(R) / f32(255), f32(G_) / f32(255), f32(B) / f32(255))
    if name == "blue":
        assert want == (29, 255, 107)        # what the reference's compiled OpenCL kernel produced
    cw, ch = 32, 16
    for s, texel in (("bgra", (B, G_, R, 255)), ("rgba", (R, G_, B, 255))):
        src = const_image(s, 20, 12, texel)
        out = run(target, cw, ch, [(f"img_{s}_{target}", src, util.full_canvas_uniforms((cw, ch), (20, 12)), 0)])
        assert np.all(out[0] == want[0]), (name, s, out[0][0, 0], want)
        if target == "nv12":
            assert np.all(out[1][..., 0] == want[1]) and np.all(out[1][..., 1] == want[2]), (name, s, out[1][0, 0], want)
        else:
            assert np.all(out[1] == want[1]) and np.all(out[2] == want[2]), (name, s)


@pytest.mark.parametrize("csc", [0, 1, 2, 3])
@pytest.mark.parametrize("yuv", [(16, 128, 128), (235, 128, 128), (81, 90, 240), (145, 54, 34), (41, 240, 110), (0, 0, 0), (255, 255, 255), (128, 77, 201)])
def test_constant_yuv_pictures_onto_a_bgra_canvas(run, yuv, csc):
    """a constant NV12 / y420p picture, any scale: every pixel is the integer matrix of its (Y, U, V)"""
    b, g, r = yuv2bgr(csc, *yuv)
    if csc == 0 and yuv == (16, 128, 128):
        assert (b, g, r) == (0, 0, 0)
    if csc == 0 and yuv == (235, 128, 128):
        assert (b, g, r) == (255, 255, 255)
    cw, ch = 48, 20
    for s, (sw, sh) in (("nv12", (72, 30)), ("y420p", (32, 50))):
        src = const_image(s, sw, sh, yuv)
        out = run("bgra", cw, ch, [(f"img_{s}_bgra", src, util.full_canvas_uniforms((cw, ch), (sw, sh)), csc)])
        assert np.all(out[0] == np.array([b, g, r, 255], dtype=np.uint8)), (s, out[0][3, 5], (b, g, r))


def box_expect(p):
    """2x2 average with the left / upper neighbour (clamped): what LINEAR filtering gives at gid - 0.5"""
    p = p.astype(np.int64)
    left = np.concatenate([p[:, :1], p[:, :-1]], axis=1)
    q = p + left
    up = np.concatenate([q[:1], q[:-1]], axis=0)
    s = q + up
    assert np.all(s % 4 == 0)
    return (s // 4).astype(np.uint8)


def test_same_size_layer_is_an_exact_half_pixel_box_filter(run):
    """power-of-two sizes make gid / size * size exact, so every tap weight is exactly 0.25; texel values are multiples of 4,
    so the average is an integer and both the unit-scale and the code-scale arithmetic must hit it exactly"""
    w, h = 64, 32
    rng = np.random.default_rng(5)
    u = util.full_canvas_uniforms((w, h), (w, h))
    # BGRA over BGRA (transform-aware family), opaque picture
    pic = util.alloc_image("bgra", w, h)
    pic[0][..., :3] = rng.integers(0, 64, (h, w, 3)) * 4
    pic[0][..., 3] = 255
    out = run("bgra", w, h, [("img_bgra_bgra_tx", pic, u, 0)])
    for c in range(3):
        assert np.array_equal(out[0][..., c], box_expect(pic[0][..., c])), c
    # NV12 over NV12 (the reference's own kernel): luma plane
    src = util.alloc_image("nv12", w, h)
    src[0][...] = rng.integers(0, 64, (h, w)) * 4
    src[1][...] = 128
    out = run("nv12", w, h, [("img_nv12_nv12", src, u, 0)])
    assert np.array_equal(out[0], box_expect(src[0]))
    # and its chroma: the owner pixel (2i, 2j) samples the half-size plane at the same normalized uv = (2i/w, 2j/h), i.e. at
    # chroma position i - 0.5: again a 2x2 box average on the chroma plane
    src[1][..., 0] = rng.integers(0, 64, (h // 2, w // 2)) * 4
    src[1][..., 1] = rng.integers(0, 64, (h // 2, w // 2)) * 4
    out = run("nv12", w, h, [("img_nv12_nv12", src, u, 0)])
    assert np.array_equal(out[1][..., 0], box_expect(src[1][..., 0])) and np.array_equal(out[1][..., 1], box_expect(src[1][..., 1]))


@pytest.mark.parametrize("opacity", [0.0, 0.25, 0.5, 0.75, 1.0])
def test_opacity_ladder_of_constants(run, opacity):
    """constant layer over a constant layer: one blend per channel, computable by hand in float32.
    BGRA target (code scale): fma(p, a, c * (1 - a)), RTE.   NV12 target (unit scale): c/255 * (1 - a) + p/255 * a, * 255, RTE."""
    cw, ch = 32, 16
    base, top = (40, 200, 90, 255), (250, 10, 130, 255)
    inv255 = f32(float.fromhex("0x1.010102p-8"))                   # RN(1/255)
    a = f32(f32(255) * f32(f32(opacity) * inv255))                 # s_A * (opacity * RN(1/255)); s_A = 255 exactly (constant alpha)
    want = []
    for c0, p in zip(base[:3], top[:3]):
        ia = f32(1) - a
        t = f32(f32(c0) * ia)
        v = np.float32(np.float64(p) * np.float64(a) + np.float64(t))      # one rounding: the fma
        want.append(int(np.clip(np.rint(v), 0, 255)))
    u0 = util.full_canvas_uniforms((cw, ch), (16, 8))
    u1 = util.full_canvas_uniforms((cw, ch), (16, 8), opacity=opacity)
    out = run("bgra", cw, ch, [("img_bgra_bgra_tx", const_image("bgra", 16, 8, base), u0, 0),
                               ("img_bgra_bgra_tx", const_image("bgra", 16, 8, top), u1, 0)])
    assert np.all(out[0][..., :3] == np.array(want, dtype=np.uint8)), (out[0][2, 2], want)
    # reference kernel, unit scale: luma of an NV12 constant over an NV12 constant
    y0, y1 = 60, 201
    al = f32(opacity)
    v = f32(f32(f32(y0) / f32(255)) * f32(f32(1) - al)) + f32(f32(f32(y1) / f32(255)) * al)
    out = run("nv12", cw, ch, [("img_nv12_nv12", const_image("nv12", 16, 8, (y0, 128, 128)), u0, 0),
                               ("img_nv12_nv12", const_image("nv12", 16, 8, (y1, 128, 128)), u1, 0)])
    assert np.all(out[0] == st8(v)), (out[0][0, 0], st8(v))
    assert np.all(out[1] == 128)


def test_metal_bgra_bgra_source_over(run):
    """kernels.metal:52-62: out.rgb = in.rgb * in.a + dst.rgb * (1 - in.a), alpha forced to 1, nearest sampling: for a
    constant source over the cleared canvas, by hand: st8(c/255 * a/255 + 0 * (1 - a/255))"""
    cw, ch = 32, 16
    src = const_image("bgra", 32, 16, (200, 100, 50, 128))
    u = util.full_canvas_uniforms((cw, ch), (32, 16))
    out = run("bgra", cw, ch, [("img_bgra_bgra", src, u, 0)])
    a = f32(128) / f32(255)
    want = [st8(f32(f32(c) / f32(255)) * a + f32(f32(0) * f32(f32(1) - a))) for c in (200, 100, 50)]
    assert np.all(out[0][..., :3] == np.array(want, dtype=np.uint8)) and np.all(out[0][..., 3] == 255)
