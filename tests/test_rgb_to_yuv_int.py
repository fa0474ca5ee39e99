"""Integer BT.601/709 RGB -> YUV (img_{bgra,rgba}_{nv12,y420p}_int): the other direction of the
north star's "bit-exact integer BT.601/709 YUV <-> RGB path".  Expectations derived here (not from oracle/ref_kernels.c):
the 16.16 coefficient tables are rebuilt from the BT.601 / BT.709 luma weights in exact rational arithmetic, pure primaries
give the published colour-bar values, and the YUV -> RGB -> YUV round trip through both integer matrices stays within a
stated bound.  Oracle leg on CPU, HIP leg with `-m gpu`."""
from fractions import Fraction as F

import numpy as np
import pytest

import util
from oracle import oracle as O
from test_first_principles import const_image, run, run_oracle, yuv2bgr  # noqa: F401  (run: fixture)

K = {0: (F(299, 1000), F(114, 1000), True), 1: (F(2126, 10000), F(722, 10000), True),
     2: (F(299, 1000), F(114, 1000), False), 3: (F(2126, 10000), F(722, 10000), False)}


def tables(csc):
    """coefficients as DESIGN.md 4.5 defines them: round(K * 65536 * range scale), green adjusted so that the luma row sums to
    round(65536 * 219/255) (65536 full range) and the chroma rows to 0"""
    kr, kb, limited = K[csc]
    kg = 1 - kr - kb
    ys, cs = (F(219, 255), F(224, 255)) if limited else (F(1), F(1))
    rnd = lambda x: int((x * 65536 + F(1, 2)).__floor__())      # noqa: E731
    y = [rnd(kr * ys), rnd(kg * ys), rnd(kb * ys)]
    u = [rnd(-kr / (2 * (1 - kb)) * cs), rnd(-kg / (2 * (1 - kb)) * cs), rnd(F(1, 2) * cs)]
    v = [rnd(F(1, 2) * cs), rnd(-kg / (2 * (1 - kr)) * cs), rnd(-kb / (2 * (1 - kr)) * cs)]
    y[1] += rnd(ys) - sum(y); u[1] -= sum(u); v[1] -= sum(v)
    return (16 if limited else 0), y, u, v


def rgb2yuv(csc, r, g, b):
    yoff, y, u, v = tables(csc)
    clip = lambda t: max(0, min(255, t >> 16))      # noqa: E731
    return (clip(y[0] * r + y[1] * g + y[2] * b + (yoff << 16) + 32768), clip(u[0] * r + u[1] * g + u[2] * b + (128 << 16) + 32768),
            clip(v[0] * r + v[1] * g + v[2] * b + (128 << 16) + 32768))


def test_tables_and_colour_bars():
    assert tables(0) == (16, [16829, 33039, 6416], [-9714, -19070, 28784], [28784, -24103, -4681])
    assert rgb2yuv(0, 255, 255, 255) == (235, 128, 128) and rgb2yuv(0, 0, 0, 0) == (16, 128, 128)
    # 100 % colour bars, BT.601 limited: the values every video engineer knows
    assert rgb2yuv(0, 255, 0, 0) == (81, 90, 240) and rgb2yuv(0, 0, 255, 0) == (145, 54, 34) and rgb2yuv(0, 0, 0, 255) == (41, 240, 110)
    assert rgb2yuv(1, 255, 255, 255) == (235, 128, 128) and rgb2yuv(2, 255, 255, 255) == (255, 128, 128) and rgb2yuv(3, 128, 128, 128)[1:] == (128, 128)
    for csc in range(4):
        for rgb in [(0, 0, 0), (255, 255, 255), (255, 0, 0), (0, 255, 0), (0, 0, 255), (17, 99, 203), (250, 3, 128)]:
            assert O.rgb2yuv_int(csc, *rgb) == rgb2yuv(csc, *rgb), (csc, rgb)


def test_oracle_matrix_exhaustive_sample():
    rng = np.random.default_rng(3)
    for csc in range(4):
        for r, g, b in rng.integers(0, 256, (4000, 3)):
            assert O.rgb2yuv_int(csc, int(r), int(g), int(b)) == rgb2yuv(csc, int(r), int(g), int(b))


@pytest.mark.parametrize("csc", [0, 1, 2, 3])
def test_round_trip_yuv_rgb_yuv(csc):
    """YUV -> RGB -> YUV (4.5) on every in-gamut code triple of a coarse lattice: each matrix rounds once, the second
    amplifies the first one's half-code error by its gain (< 1): |dY| <= 1, |dU|, |dV| <= 1 wherever the RGB value did not clip"""
    lim = csc < 2
    ys = range(16, 236, 7) if lim else range(0, 256, 8)
    cs = range(16, 241, 8) if lim else range(0, 256, 9)
    worst = [0, 0, 0]
    n = 0
    for y in ys:
        for u in cs:
            for v in cs:
                b, g, r = yuv2bgr(csc, y, u, v)
                if min(r, g, b) == 0 or max(r, g, b) == 255:
                    continue                          # out of the RGB gamut: clipped, no round trip to expect
                y2, u2, v2 = rgb2yuv(csc, r, g, b)
                for k, d in enumerate((abs(y2 - y), abs(u2 - u), abs(v2 - v))):
                    worst[k] = max(worst[k], d)
                n += 1
    assert n > 2000 and worst[0] <= 1 and worst[1] <= 1 and worst[2] <= 1, (n, worst)


@pytest.mark.parametrize("csc", [0, 1, 2, 3])
@pytest.mark.parametrize("target", ["nv12", "y420p"])
def test_constant_pictures_through_the_kernels(run, target, csc):
    """an opaque constant BGRA / RGBA picture over the whole canvas, any scale: every pixel is the integer matrix of its colour"""
    cw, ch = 48, 20
    for rgb in [(255, 255, 255), (0, 0, 0), (255, 0, 0), (0, 255, 0), (0, 0, 255), (17, 99, 203)]:
        want = rgb2yuv(csc, *rgb)
        for s, texel in (("bgra", (rgb[2], rgb[1], rgb[0], 255)), ("rgba", (rgb[0], rgb[1], rgb[2], 255))):
            src = const_image(s, 30, 14, texel)
            out = run(target, cw, ch, [(f"img_{s}_{target}_int", src, util.full_canvas_uniforms((cw, ch), (30, 14)), csc)])
            assert np.all(out[0] == want[0]), (rgb, s, out[0][0, 0], want)
            got_u, got_v = (out[1][..., 0], out[1][..., 1]) if target == "nv12" else (out[1], out[2])
            assert np.all(got_u == want[1]) and np.all(got_v == want[2]), (rgb, s, got_u[0, 0], got_v[0, 0], want)


def test_translucent_constant_over_constant(run):
    """blend on the code scale: r = fma(P, a, cur * (1 - a)) with a = 128 * (opacity * RN(1/255)), one rounding, then RTE"""
    f32 = np.float32
    cw, ch, opacity = 32, 16, 0.75
    base, top = (200, 40, 90, 255), (16, 240, 128, 128)        # B, G, R, A
    p0 = rgb2yuv(0, base[2], base[1], base[0])
    p1 = rgb2yuv(0, top[2], top[1], top[0])
    a = f32(f32(128) * f32(f32(opacity) * f32(float.fromhex("0x1.010102p-8"))))
    want = [int(np.rint(np.float32(np.float64(q) * np.float64(a) + np.float64(f32(f32(c) * f32(f32(1) - a)))))) for c, q in zip(p0, p1)]
    u0 = util.full_canvas_uniforms((cw, ch), (16, 8))
    u1 = util.full_canvas_uniforms((cw, ch), (16, 8), opacity=opacity)
    out = run("nv12", cw, ch, [("img_bgra_nv12_int", const_image("bgra", 16, 8, base), u0, 0),
                               ("img_bgra_nv12_int", const_image("bgra", 16, 8, top), u1, 0)])
    assert np.all(out[0] == want[0]) and np.all(out[1][..., 0] == want[1]) and np.all(out[1][..., 1] == want[2]), (out[0][0, 0], out[1][0, 0], want)
