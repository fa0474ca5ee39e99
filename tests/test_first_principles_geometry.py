"""More vectors whose expected bytes are derived in the test itself (see test_first_principles.py) — this file covers what
that one leaves to the oracle: NON-TRIVIAL filter weights and tap positions, border / fill scenes, per-pixel alpha through the
RGB -> YUV blend, chroma ownership on odd-sized 4:2:0 canvases, Lanczos-3.  Every expectation is computed here from the
reference's kernel text (kernels.cl.swift, line numbers below) or from DESIGN.md section 4 with exact rational / explicit
float32 arithmetic; nothing comes from oracle/ref_kernels.c.  Each test runs on the oracle (CPU leg) and on the HIP path
(`-m gpu` leg).

How exactness is forced.  Canvas sizes are powers of two, so out_uv = gid / size and every product of the geometry prologue
(kernels.cl.swift:70-77) are exact in float32; the layer rectangles have power-of-two sizes, so the inverse matrices are
exact too.  The LINEAR sampler (OpenCL 1.2 section 8.2) then sees positions u * w - 0.5 that are exact dyadic rationals:
  * 3:2 reduction (source = 1.5 x canvas):  1.5 gid - 0.5  ->  even gid: taps (3k - 1, 3k), a = 1/2;  odd gid: tap 3k + 1, a = 0
  * 4:1 reduction:  4 gid - 0.5  ->  taps (4 gid - 1, 4 gid), a = 1/2
  * chroma of a 3:2 NV12 -> BGRA conversion (half-size planes at the same uv):  0.75 gid - 0.5  ->  a in {1/2, 1/4, 0, 3/4}
Texel values are multiples of 16, so every weighted sum is an integer: the code-scale family must produce it
exactly, and the reference's unit-scale arithmetic (c / 255 per tap, * 255 at the store) lands within 1e-4 of it, i.e. on it
after the store's rounding."""
import numpy as np
import pytest
from fractions import Fraction

import util
from oracle import oracle as O
from test_first_principles import f32, st8, yuv2bgr, rgb2yuv_codes, const_image, run, run_oracle  # noqa: F401  (run: fixture)


# ---------------------------------------------------------------------------------------------------------------
# exact LINEAR / CLAMP_TO_EDGE sampling at a dyadic position (Fractions; OpenCL 1.2 section 8.2)
# ---------------------------------------------------------------------------------------------------------------
def lin_taps(pos, n):
    """pos = u * w - 0.5 as a Fraction -> [(index, weight)], indices clamped to [0, n - 1]"""
    i0 = pos.numerator // pos.denominator          # floor
    a = pos - i0
    return [(min(max(i0, 0), n - 1), 1 - a), (min(max(i0 + 1, 0), n - 1), a)]


def sample_exact(plane, px, py):
    """bilinear sample of an integer plane at (px, py) (Fractions, texel units, already minus 0.5): an exact Fraction"""
    h, w = plane.shape
    acc = Fraction(0)
    for j, wj in lin_taps(py, h):
        for i, wi in lin_taps(px, w):
            acc += wi * wj * int(plane[j, i])
    return acc


def expect_plane(plane, cw, ch, scale_x, scale_y):
    """full-canvas layer: canvas pixel (x, y) samples at (x * scale_x - 1/2, y * scale_y - 1/2)"""
    out = np.zeros((ch, cw), dtype=np.int64)
    for y in range(ch):
        for x in range(cw):
            v = sample_exact(plane, Fraction(x) * scale_x - Fraction(1, 2), Fraction(y) * scale_y - Fraction(1, 2))
            assert v.denominator == 1, "test vector must make every sample an integer"
            out[y, x] = int(v)
    return out


def mult16(rng, shape):
    return (rng.integers(0, 16, shape) * 16).astype(np.uint8)


REDUCTIONS = {"3:2": (Fraction(3, 2), 96, 48), "4:1": (Fraction(4), 256, 128)}
CW, CH = 64, 32


@pytest.mark.parametrize("ratio", list(REDUCTIONS))
def test_exact_reductions_bgra_over_bgra(run, ratio):
    """img_bgra_bgra_tx / img_rgba_bgra_tx, opaque texels: every channel is the exact weighted tap sum (weights 1/4, 1/2, 1)"""
    s, sw, sh = REDUCTIONS[ratio]
    rng = np.random.default_rng(11)
    u = util.full_canvas_uniforms((CW, CH), (sw, sh))
    for fmt, order in (("bgra", (0, 1, 2)), ("rgba", (2, 1, 0))):
        pic = util.alloc_image(fmt, sw, sh)
        pic[0][..., :3] = mult16(rng, (sh, sw, 3))
        pic[0][..., 3] = 255
        out = run("bgra", CW, CH, [(f"img_{fmt}_bgra_tx", pic, u, 0)])
        for c in range(3):          # canvas channel c (B, G, R) comes from source channel order[c]
            assert np.array_equal(out[0][..., c], expect_plane(pic[0][..., order[c]], CW, CH, s, s)), (fmt, ratio, c)
        assert np.all(out[0][..., 3] == 255)


@pytest.mark.parametrize("ratio", list(REDUCTIONS))
@pytest.mark.parametrize("fmt", ["nv12", "y420p"])
def test_exact_reductions_yuv_to_bgra(run, ratio, fmt):
    """img_nv12_bgra / img_y420p_bgra: luma through the full-range BT.601 matrix with neutral chroma is the identity
    (R = G = B = (65536 Y + 32768) >> 16 = Y), so the canvas shows the exact luma sample; then varying chroma with constant luma:
    the canvas is the integer matrix of (Y, exact U sample, exact V sample) — chroma weights 1/4, 1/2, 3/4, 1 at 3:2"""
    s, sw, sh = REDUCTIONS[ratio]
    rng = np.random.default_rng(12)
    u = util.full_canvas_uniforms((CW, CH), (sw, sh))
    src = util.alloc_image(fmt, sw, sh)
    src[0][...] = mult16(rng, (sh, sw))
    for p in src[1:]:
        p[...] = 128
    out = run("bgra", CW, CH, [(f"img_{fmt}_bgra", src, u, 2)])
    want = expect_plane(src[0], CW, CH, s, s)
    for c in range(3):
        assert np.array_equal(out[0][..., c], want), (fmt, ratio, c)
    # chroma: half-size planes sampled at the same normalized uv = gid / size  ->  gid * s / 2 - 1/2 in chroma texels
    src[0][...] = 120
    if fmt == "nv12":
        src[1][..., 0] = mult16(rng, (sh // 2, sw // 2)); src[1][..., 1] = mult16(rng, (sh // 2, sw // 2))
        up, vp = src[1][..., 0], src[1][..., 1]
    else:
        src[1][...] = mult16(rng, (sh // 2, sw // 2)); src[2][...] = mult16(rng, (sh // 2, sw // 2))
        up, vp = src[1], src[2]
    eu, ev = expect_plane(up, CW, CH, s / 2, s / 2), expect_plane(vp, CW, CH, s / 2, s / 2)
    for csc in (0, 3):
        out = run("bgra", CW, CH, [(f"img_{fmt}_bgra", src, u, csc)])
        want = np.array([[yuv2bgr(csc, 120, int(eu[y, x]), int(ev[y, x])) for x in range(CW)] for y in range(CH)], dtype=np.uint8)
        assert np.array_equal(out[0][..., :3], want), (fmt, ratio, csc)


@pytest.mark.parametrize("ratio", list(REDUCTIONS))
@pytest.mark.parametrize("kernel", ["img_nv12_nv12", "img_y420p_y420p", "img_y420p_nv12"])
def test_exact_reductions_reference_yuv_kernels(run, ratio, kernel):
    """the reference's own kernels (kernels.cl.swift:47-109, 186-255, 267-335), opaque: luma at every pixel, chroma at the quad's
    even/even pixel (`handleChroma`, :76) sampled on the half-size plane at that pixel's uv: 2 i s / 2 - 1/2 = i s - 1/2"""
    s, sw, sh = REDUCTIONS[ratio]
    rng = np.random.default_rng(13)
    sfmt, tfmt = kernel.split("_")[1], kernel.split("_")[2]
    src = util.alloc_image(sfmt, sw, sh)
    src[0][...] = mult16(rng, (sh, sw))
    if sfmt == "nv12":
        src[1][..., 0] = mult16(rng, (sh // 2, sw // 2)); src[1][..., 1] = mult16(rng, (sh // 2, sw // 2))
        up, vp = src[1][..., 0], src[1][..., 1]
    else:
        src[1][...] = mult16(rng, (sh // 2, sw // 2)); src[2][...] = mult16(rng, (sh // 2, sw // 2))
        up, vp = src[1], src[2]
    out = run(tfmt, CW, CH, [(kernel, src, util.full_canvas_uniforms((CW, CH), (sw, sh)), 0)])
    assert np.array_equal(out[0], expect_plane(src[0], CW, CH, s, s)), (kernel, ratio, "luma")
    eu, ev = expect_plane(up, CW // 2, CH // 2, s, s), expect_plane(vp, CW // 2, CH // 2, s, s)
    got_u, got_v = (out[1][..., 0], out[1][..., 1]) if tfmt == "nv12" else (out[1], out[2])
    assert np.array_equal(got_u, eu) and np.array_equal(got_v, ev), (kernel, ratio, "chroma")


@pytest.mark.parametrize("ratio", list(REDUCTIONS))
@pytest.mark.parametrize("kernel", ["img_bgra_nv12", "img_rgba_y420p"])
def test_exact_reductions_rgb_to_yuv(run, ratio, kernel):
    """img_{bgra,rgba}_{nv12,y420p} (kernels.cl.swift:469-532), opaque texels, no fill: the sampled (r, g, b) are exact dyadic sums
    / 255 up to the sampler's rounding, far inside the margin of the store, so Y, U, V = rgb2yuv of the exact samples — evaluated
    here in float32 in the kernel's order (vecmat4 = dot, summed left to right)"""
    s, sw, sh = REDUCTIONS[ratio]
    rng = np.random.default_rng(14)
    sfmt, tfmt = kernel.split("_")[1], kernel.split("_")[2]
    pic = util.alloc_image(sfmt, sw, sh)
    pic[0][..., :3] = mult16(rng, (sh, sw, 3))
    pic[0][..., 3] = 255
    out = run(tfmt, CW, CH, [(kernel, pic, util.full_canvas_uniforms((CW, CH), (sw, sh)), 0)])
    ri, gi, bi = (0, 1, 2) if sfmt == "rgba" else (2, 1, 0)
    er, eg, eb = (expect_plane(pic[0][..., k], CW, CH, s, s) for k in (ri, gi, bi))
    # Expected codes from the EXACT samples in rational arithmetic (coefficients = the float32 constants of the kernel text): the
    # kernel's float32 evaluation differs from the exact value by < 1e-5 codes, so every pixel whose exact value is at least 1e-3
    # codes away from a rounding tie (x.5) is forced; the few that sit on a tie (r = g = b gives U = V = 127.5 exactly) are skipped
    rows = [[Fraction(float(f32(c))) for c in m] for m in ((0.299, 0.587, 0.113, 0.0), (-0.169, -0.331, 0.5, 0.5), (0.5, -0.419, -0.081, 0.5))]
    got_u, got_v = (out[1][..., 0], out[1][..., 1]) if tfmt == "nv12" else (out[1], out[2])
    checked = skipped = 0
    for y in range(CH):
        for x in range(CW):
            rgb = [Fraction(int(e[y, x]), 255) for e in (er, eg, eb)]
            for comp, m in enumerate(rows):
                if comp > 0 and (x % 2 or y % 2):
                    continue
                exact = (rgb[0] * m[0] + rgb[1] * m[1] + rgb[2] * m[2] + m[3]) * 255
                if abs((exact % 1) - Fraction(1, 2)) < Fraction(1, 1000):
                    skipped += 1
                    continue
                want = min(max(int(exact + Fraction(1, 2)), 0), 255)           # round to nearest, away from the excluded ties
                got = int(out[0][y, x]) if comp == 0 else int((got_u if comp == 1 else got_v)[y // 2, x // 2])
                assert got == want, (kernel, ratio, x, y, comp, got, want)
                checked += 1
    assert checked > 0.95 * (CW * CH * 3 // 2) and skipped < 0.05 * (CW * CH * 3 // 2), (checked, skipped)


# ---------------------------------------------------------------------------------------------------------------
# border + fill scenes on a constant canvas
# ---------------------------------------------------------------------------------------------------------------
# canvas 128 x 64, picture rectangle (48, 24, 32, 16), border 16 / 8 on every side: all matrix entries are exact, and the
# inclusive tests of the kernels (`>= 0 && <= 1`, kernels.cl.swift:77,84) select exactly
#   border quad: x in [32, 96], y in [16, 48]      picture: x in [48, 80], y in [24, 40]      (both ends inclusive)
BW, BH = 128, 64
RECT, BORDER = (48, 24, 32, 16), (16, 8, 16, 8)
in_border = lambda x, y: 32 <= x <= 96 and 16 <= y <= 48      # noqa: E731
in_pict = lambda x, y: 48 <= x <= 80 and 24 <= y <= 40        # noqa: E731


def fill_yuv(fill, scale):
    """vecmat4((fill.rgb * scale, 1), rgb2yuv): float32, dot summed left to right (kernels.cl.swift:96-102 / :509-510)"""
    rows = [(0.299, 0.587, 0.113, 0.0), (-0.169, -0.331, 0.5, 0.5), (0.5, -0.419, -0.081, 0.5)]
    r, g, b = (f32(f32(c) * f32(scale)) for c in fill[:3])
    out = []
    for m in rows:
        acc = f32(r * f32(m[0])) + f32(g * f32(m[1]))
        acc = f32(acc) + f32(b * f32(m[2]))
        acc = f32(acc) + f32(f32(1.0) * f32(m[3]))
        out.append(f32(acc))
    return out


def clampf(v, lo, hi):
    return f32(min(max(f32(v), f32(lo)), f32(hi)))


@pytest.mark.parametrize("target", ["nv12", "y420p"])
def test_border_and_fill_yuv_source_paints_the_border(run, target):
    """A YUV-source kernel on a constant canvas with a constant picture: outside the border quad nothing changes; inside it but
    outside the picture the FILL is blended (alpha = opacity * fill.w, luma clamped to [0,1], chroma to [-1,1]); inside the picture
    the sample is blended by the opacity alone (kernels.cl.swift:78-105)."""
    canvas_yuv, pict_yuv = (60, 100, 180), (200, 40, 220)
    fill, opacity = (0.25, 0.5, 0.75, 0.5), 0.75
    k = f"img_{target}_{target}"
    base = const_image(target, BW, BH, canvas_yuv)
    pic = const_image(target, 32, 16, pict_yuv)
    u0 = util.full_canvas_uniforms((BW, BH), (BW, BH))
    u1 = util.make_uniforms((BW, BH), rect=RECT, border=BORDER, fill=fill, opacity=opacity, in_size=(32, 16))
    out = run(target, BW, BH, [(k, base, u0, 0), (k, pic, u1, 0)])
    al_f = f32(f32(opacity) * f32(fill[3]))
    fy, fu, fv = fill_yuv(fill, 1.0)
    one = f32(1)

    def blend(cur, new, a, lo, hi, clamp):
        v = f32(f32(f32(cur) / f32(255)) * f32(one - a)) + f32(f32(new) * a)
        return st8(clampf(v, lo, hi) if clamp else v)

    want_y = np.zeros((BH, BW), dtype=np.uint8)
    want_u = np.zeros((BH // 2, BW // 2), dtype=np.uint8)
    want_v = np.zeros((BH // 2, BW // 2), dtype=np.uint8)
    for y in range(BH):
        for x in range(BW):
            if in_pict(x, y):
                vals = [blend(c, f32(p) / f32(255), f32(opacity), 0, 1, False) for c, p in zip(canvas_yuv, pict_yuv)]
            elif in_border(x, y):
                vals = [blend(canvas_yuv[0], fy, al_f, 0, 1, True), blend(canvas_yuv[1], fu, al_f, -1, 1, True), blend(canvas_yuv[2], fv, al_f, -1, 1, True)]
            else:
                vals = list(canvas_yuv)
            want_y[y, x] = vals[0]
            if x % 2 == 0 and y % 2 == 0:
                want_u[y // 2, x // 2], want_v[y // 2, x // 2] = vals[1], vals[2]
    assert np.array_equal(out[0], want_y)
    got_u, got_v = (out[1][..., 0], out[1][..., 1]) if target == "nv12" else (out[1], out[2])
    assert np.array_equal(got_u, want_u) and np.array_equal(got_v, want_v)


@pytest.mark.parametrize("alpha", [255, 128])
def test_border_and_fill_rgb_source_keeps_the_border(run, alpha):
    """An RGB-source kernel paints NOTHING between border quad and picture (its writes sit inside the `tx` test,
    kernels.cl.swift:509-529); inside the picture the fill colour is pre-multiplied by alpha = opacity * fill.w AND blended with
    it (:509-513), then the pixel, pre-multiplied by its own alpha x opacity, is blended on top (:516-521) — per-pixel alpha
    included (alpha = 128: every blend weight is a genuine fraction)."""
    canvas_yuv, texel = (60, 100, 180), (240, 16, 80, alpha)        # B, G, R, A
    fill, opacity = (0.25, 0.5, 0.75, 0.5), 0.75
    base = const_image("nv12", BW, BH, canvas_yuv)
    pic = const_image("bgra", 32, 16, texel)
    u0 = util.full_canvas_uniforms((BW, BH), (BW, BH))
    u1 = util.make_uniforms((BW, BH), rect=RECT, border=BORDER, fill=fill, opacity=opacity, in_size=(32, 16))
    out = run("nv12", BW, BH, [("img_nv12_nv12", base, u0, 0), ("img_bgra_nv12", pic, u1, 0)])
    one = f32(1)
    af = f32(f32(opacity) * f32(fill[3]))
    fy, fu, fv = fill_yuv(fill, af)                                  # fill.rgb * alpha, then the matrix
    cur = [f32(f32(c) / f32(255)) for c in canvas_yuv]
    r0 = f32(f32(cur[0] * f32(one - af)) + f32(fy * af))
    r1 = clampf(f32(f32(cur[1] * f32(one - af)) + f32(fu * af)), -1, 1)
    r2 = clampf(f32(f32(cur[2] * f32(one - af)) + f32(fv * af)), -1, 1)
    # the sampled texel: constant picture -> (B, G, R, A) / 255 exactly as stored (a convex combination of equal values)
    b, g, r, a = (f32(f32(c) / f32(255)) for c in texel)
    a2 = f32(a * f32(opacity))
    rows = [(0.299, 0.587, 0.113, 0.0), (-0.169, -0.331, 0.5, 0.5), (0.5, -0.419, -0.081, 0.5)]
    yuv = []
    for m in rows:
        acc = f32(f32(r * a2) * f32(m[0])) + f32(f32(g * a2) * f32(m[1]))
        acc = f32(acc) + f32(f32(b * a2) * f32(m[2]))
        acc = f32(acc) + f32(one * f32(m[3]))
        yuv.append(f32(acc))
    res = [st8(f32(f32(rc * f32(one - a2)) + f32(yc * a2))) for rc, yc in zip((r0, r1, r2), yuv)]
    want_y = np.full((BH, BW), canvas_yuv[0], dtype=np.uint8)
    want_u = np.full((BH // 2, BW // 2), canvas_yuv[1], dtype=np.uint8)
    want_v = np.full((BH // 2, BW // 2), canvas_yuv[2], dtype=np.uint8)
    for y in range(BH):
        for x in range(BW):
            if in_pict(x, y):
                want_y[y, x] = res[0]
                if x % 2 == 0 and y % 2 == 0:
                    want_u[y // 2, x // 2], want_v[y // 2, x // 2] = res[1], res[2]
    assert np.array_equal(out[0], want_y)
    assert np.array_equal(out[1][..., 0], want_u) and np.array_equal(out[1][..., 1], want_v)


@pytest.mark.parametrize("src", ["nv12", "bgra"])
def test_border_and_fill_on_a_bgra_canvas(run, src):
    """The BGRA-target family (spec owned by this repository): BOTH source classes paint the fill inside the border
    quad — r = clamp(fma(fill * 255, af, cur * (1 - af)), 0, 255) — and blend the picture on top of it inside the picture:
    r = fma(p, a, r * (1 - a)), a = opacity (YUV sources) or sample alpha * (opacity * RN(1/255)) (RGB sources); RTE store."""
    canvas, fill, opacity = (40, 200, 90, 255), (0.25, 0.5, 0.75, 0.5), 0.75
    u0 = util.full_canvas_uniforms((BW, BH), (BW, BH))
    u1 = util.make_uniforms((BW, BH), rect=RECT, border=BORDER, fill=fill, opacity=opacity, in_size=(32, 16))
    base = const_image("bgra", BW, BH, canvas)
    if src == "nv12":
        pic, k = const_image("nv12", 32, 16, (145, 54, 34)), "img_nv12_bgra"
        p = [f32(c) for c in yuv2bgr(0, 145, 54, 34)]
        a = f32(f32(1.0) * f32(opacity))
    else:
        pic, k = const_image("bgra", 32, 16, (250, 10, 130, 128)), "img_bgra_bgra_tx"
        p = [f32(250), f32(10), f32(130)]
        a = f32(f32(128) * f32(f32(opacity) * f32(float.fromhex("0x1.010102p-8"))))
    fma = lambda x, y, z: np.float32(np.float64(x) * np.float64(y) + np.float64(z))      # noqa: E731  (one rounding)
    af = f32(f32(opacity) * f32(fill[3]))
    fcode = [f32(f32(fill[2]) * f32(255)), f32(f32(fill[1]) * f32(255)), f32(f32(fill[0]) * f32(255))]     # memory order B, G, R
    rfill = [clampf(fma(fc, af, f32(f32(c) * f32(f32(1) - af))), 0, 255) for fc, c in zip(fcode, canvas[:3])]
    rpict = [fma(pc, a, f32(rc * f32(f32(1) - a))) for pc, rc in zip(p, rfill)]
    code = lambda v: int(np.clip(np.rint(v), 0, 255))      # noqa: E731
    out = run("bgra", BW, BH, [("img_bgra_bgra_tx", base, u0, 0), (k, pic, u1, 0)])
    want = np.zeros((BH, BW, 4), dtype=np.uint8)
    for y in range(BH):
        for x in range(BW):
            v = [code(c) for c in rpict] if in_pict(x, y) else [code(c) for c in rfill] if in_border(x, y) else list(canvas[:3])
            want[y, x] = v + [255]
    assert np.array_equal(out[0], want), (src, out[0][20, 40], want[20, 40], out[0][30, 60], want[30, 60])


# ---------------------------------------------------------------------------------------------------------------
# chroma ownership on an odd-sized 4:2:0 canvas
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("target", ["nv12", "y420p"])
def test_chroma_ownership_on_a_7x5_canvas(run, target):
    """7 x 5 luma, 3 x 2 chroma.  The quad's chroma belongs to its even/even pixel (kernels.cl.swift:76); a chroma sample (i, j) is
    written iff pixel (2 i, 2 j) is inside the picture; gid / 2 of the last even column / row (x = 6, y = 4) lies outside the
    chroma plane: nothing is written there and nothing is read from beyond it.  The layer's rectangle
    (0.5, 0.5) .. (5.5, 4.5) keeps every inside / outside decision half a pixel away from its boundary."""
    cw, ch = 7, 5
    k = f"img_{target}_{target}"
    pic = const_image(target, 8, 8, (200, 40, 220))
    u = util.make_uniforms((cw, ch), rect=(0.5, 0.5, 5.0, 4.0), in_size=(8, 8))
    out = run(target, cw, ch, [(k, pic, u, 0)])                                   # on the cleared canvas: Y = 0, chroma = 128
    inside = lambda x, y: 1 <= x <= 5 and 1 <= y <= 4      # noqa: E731
    want_y = np.array([[200 if inside(x, y) else 0 for x in range(cw)] for y in range(ch)], dtype=np.uint8)
    assert np.array_equal(out[0], want_y)
    want_u = np.array([[40 if inside(2 * i, 2 * j) else 128 for i in range(3)] for j in range(2)], dtype=np.uint8)
    want_v = np.array([[220 if inside(2 * i, 2 * j) else 128 for i in range(3)] for j in range(2)], dtype=np.uint8)
    got_u, got_v = (out[1][..., 0], out[1][..., 1]) if target == "nv12" else (out[1], out[2])
    assert np.array_equal(got_u, want_u) and np.array_equal(got_v, want_v), (got_u, want_u)


# ---------------------------------------------------------------------------------------------------------------
# Lanczos-3: properties that hold for any correct separable filter with normalised weights
# ---------------------------------------------------------------------------------------------------------------
def lanczos(ctx, src, ow, oh):
    """through the oracle (ctx is None) or the HIP path"""
    ih, iw = src.shape[:2]
    if ctx is None:
        dst = util.alloc_image("bgra", ow, oh)
        assert O.lanczos_bgra(dst[0], src) == 0
        return dst[0]
    import gpuutil as G
    from swiftvideo_amd import compute as sv
    gs = G.to_gpu(ctx, "bgra", iw, ih, [src])
    gd = G.to_gpu(ctx, "bgra", ow, oh, util.alloc_image("bgra", ow, oh))
    sv.usingContext(ctx, lambda c: sv.scaleLanczos(c, gd, gs))
    return G.from_gpu(ctx, gd, "bgra", ow, oh)[0]


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def lz(request):
    ctx = None if request.param == "oracle" else request.getfixturevalue("ctx")
    return lambda src, ow, oh: lanczos(ctx, src, ow, oh)


@pytest.mark.parametrize("iw,ih,ow,oh", [(96, 54, 48, 27), (64, 36, 128, 72), (200, 120, 75, 45), (33, 17, 20, 10)])
def test_lanczos_constant_in_constant_out(lz, iw, ih, ow, oh):
    """weights sum to 1 (normalised in double, rounded to float32: the sum is 1 +- 1e-6), so a constant picture stays constant"""
    for texel in ((0, 0, 0, 0), (255, 255, 255, 255), (13, 77, 200, 128)):
        src = util.alloc_image("bgra", iw, ih)[0]
        src[...] = np.array(texel, dtype=np.uint8)
        out = lz(src, ow, oh)
        assert np.all(out == np.array(texel, dtype=np.uint8)), (texel, out[0, 0])


def test_lanczos_step_is_symmetric_and_rings_within_bounds(lz):
    """a vertical step edge at the centre, 2:1 reduction: far from the edge the constants survive exactly; the response is
    point-symmetric about the edge (out[x] + out[W - 1 - x] = a + b within the two roundings); the Lanczos-3 ringing stays
    within 10 % of the step (first side lobe of the kernel: < 9.1 %)"""
    iw, ih, ow, oh, a, b = 128, 8, 64, 4, 40, 200
    src = util.alloc_image("bgra", iw, ih)[0]
    src[:, : iw // 2] = a
    src[:, iw // 2:] = b
    out = lz(src, ow, oh).astype(np.int64)
    assert np.all(out[:, :20] == a) and np.all(out[:, -20:] == b)
    assert np.all(np.abs(out + out[:, ::-1] - (a + b)) <= 1)
    assert out.min() >= a - 0.1 * (b - a) and out.max() <= b + 0.1 * (b - a)
    assert np.all(out == out[0:1])                        # rows are identical: the vertical pass sees constants
