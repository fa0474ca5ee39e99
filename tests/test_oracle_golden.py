"""The oracle against the committed golden vectors, against the compiled reference kernel
strings (when oracle/_ref/libclref.so is available), and its primitives against their
definitions.  CPU only."""
from fractions import Fraction
from pathlib import Path

import zlib

import numpy as np
import pytest

import scenarios as S
import util
from oracle import oracle as O

GOLD = Path(__file__).resolve().parent / "golden" / "vectors.npz"


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def _outs(gold, key):
    outs, i = [], 0
    while f"{key}/out{i}" in gold:
        outs.append(gold[f"{key}/out{i}"])
        i += 1
    return outs


def test_golden_file_covers_every_kernel_and_scenario(gold):
    index = set(gold["index"].tolist())
    for k in S.LAYER_KERNELS_REF + S.LAYER_KERNELS_OWN + ["img_bgra_bgra"]:
        for sc in S.SCENARIOS:
            assert f"{k}/{sc}" in index
    assert len(index) >= 117


def test_oracle_reproduces_golden_vectors(gold):
    for key in gold["index"].tolist():
        kernel = key.split("/")[0]
        cw, ch, iw, ih, seed = gold[key + "/meta"].tolist()
        exp = _outs(gold, key)
        if kernel == "lanczos3":
            src = util.alloc_image("bgra", iw, ih, seed=seed)
            dst = util.alloc_image("bgra", cw, ch)
            assert O.lanczos_bgra(dst[0], src[0], threads=2) == 0
            got = dst
        elif kernel.startswith("img_clear"):
            got = util.alloc_image(kernel.split("_")[2], cw, ch, seed=seed)
            assert O.run_kernel(kernel, got) == 0
        else:
            _, s, d = kernel.split("_")[:3]
            src = util.alloc_image(s, iw, ih, seed=seed)
            got = util.alloc_image(d, cw, ch, seed=seed + 1000)
            assert O.run_kernel(kernel, got, src, gold[key + "/uniforms"], threads=3) == 0
        for g, e in zip(got, exp):
            assert np.array_equal(g, e), key


def test_oracle_equals_compiled_reference_kernels():
    """Kernel-body arithmetic of the restatement == the reference's OpenCL-C source compiled for
    x86-64 (a cross-check, not a reference build: the image builtins are ours, oracle/clref/cl_shim.c)."""
    if O.clref() is None:
        pytest.skip("oracle/_ref/libclref.so not built (needs /root/reference)")
    rng_geo = [dict(rect=(3, 1, 50, 30), rotation=0.9, opacity=0.45, fill=(0.3, 0.9, 0.2, 0.65), border=(4, 0, 1, 6)),
               dict(rect=(-30, -10, 140, 70), opacity=1.0, tex=(0.3, 0.4, 0.3, 0.2)),
               dict(rect=(40, 20, 10, 8), opacity=0.2, fill=(1, 1, 1, 1), border=(30, 15, 10, 5))]
    for kernel in S.LAYER_KERNELS_REF:
        _, s, d = kernel.split("_")
        for gi, geo in enumerate(rng_geo):
            u = util.make_uniforms((64, 36), in_size=(48, 28), **geo)
            src = util.alloc_image(s, 48, 28, seed=300 + gi)
            c0 = util.alloc_image(d, 64, 36, seed=400 + gi)
            a, b = util.copy_image(c0), util.copy_image(c0)
            assert O.run_kernel(kernel, a, src, u) == 0
            assert O.run_clref(kernel, b, src, u) == 0
            for x, y in zip(a, b):
                assert np.array_equal(x, y), f"{kernel} geometry {gi}"


def test_threaded_oracle_equals_single_thread():
    u = util.make_uniforms((130, 47), rect=(5, 3, 100, 40), rotation=0.1, opacity=0.7, in_size=(64, 36))
    for kernel in ["img_bgra_nv12", "img_y420p_y420p", "img_nv12_bgra"]:
        _, s, d = kernel.split("_")
        src = util.alloc_image(s, 64, 36, seed=8)
        c0 = util.alloc_image(d, 130, 47, seed=9)
        a, b = util.copy_image(c0), util.copy_image(c0)
        assert O.run_kernel(kernel, a, src, u, threads=1) == 0
        assert O.run_kernel(kernel, b, src, u, threads=7) == 0
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


# ---- primitives -------------------------------------------------------------------------------
def test_unorm8_load_is_correctly_rounded_division():
    lib = O.lib()
    for c in range(256):
        got = np.float32(lib.orc_load_unorm8(c))
        assert got == np.float32(c) / np.float32(255)
        # correctly rounded: the exact quotient lies within half an ulp
        exact = Fraction(c, 255)
        ulp = Fraction(float(np.spacing(got))) if c else Fraction(0)
        assert abs(Fraction(float(got)) - exact) <= ulp / 2


def test_two_op_unorm8_formula_is_exact():
    """pixel_math.hip.h computes c/255 as fma(c, r_hi, c*r_lo); prove it in exact arithmetic."""
    def rn32(x):
        return Fraction(float(np.float32(float(x)))) if x.denominator.bit_length() < 60 else None
    r_hi = Fraction(float(np.float32(1) / np.float32(255)))
    r_lo = Fraction(float.fromhex("-0x1.fdfdfep-33"))
    for c in range(256):
        t = Fraction(float(np.float32(float(c * r_lo))))          # c*r_lo is exactly representable in double
        s = c * r_hi + t                                            # exact
        # round the exact sum to float32: both terms are dyadic, double holds the sum exactly here
        assert np.float32(float(s)) == np.float32(c) / np.float32(255)


def test_unorm8_store_rounding():
    lib = O.lib()
    assert lib.orc_store_unorm8(0.5) == 128            # 127.5 -> even
    assert lib.orc_store_unorm8(0.0) == 0 and lib.orc_store_unorm8(1.0) == 255
    assert lib.orc_store_unorm8(-3.0) == 0 and lib.orc_store_unorm8(7.0) == 255
    assert lib.orc_store_unorm8(float("nan")) == 0
    assert lib.orc_store_unorm8(float("inf")) == 255 and lib.orc_store_unorm8(float("-inf")) == 0
    assert lib.orc_store_unorm8(2.5 / 255) == 2 and lib.orc_store_unorm8(3.5 / 255) == 4
    for c in range(256):                                 # round trip: store(load(c)) == c
        assert lib.orc_store_unorm8(lib.orc_load_unorm8(c)) == c


def test_integer_yuv2rgb_known_answers():
    # BT.601 limited: black, white, and the primaries' well-known code points
    assert O.yuv2rgb_int(0, 16, 128, 128) == (0, 0, 0)
    assert O.yuv2rgb_int(0, 235, 128, 128) == (255, 255, 255)
    # the 8-bit code points of the primaries come back within one code
    for yuv, rgb in (((81, 90, 240), (255, 0, 0)), ((145, 54, 34), (0, 255, 0)), ((41, 240, 110), (0, 0, 255))):
        got = O.yuv2rgb_int(0, *yuv)
        assert all(abs(g - e) <= 1 for g, e in zip(got, rgb)), (yuv, got)
    assert O.yuv2rgb_int(0, 0, 0, 0) == (0, 136, 0)       # negative sums: arithmetic shift, then clip
    # full range: identity on grey
    for y in (0, 1, 127, 128, 254, 255):
        assert O.yuv2rgb_int(2, y, 128, 128) == (y, y, y)
        assert O.yuv2rgb_int(3, y, 128, 128) == (y, y, y)
    # against a float64 evaluation of the same fixed-point definition, all (y,u,v) on a grid
    k = {0: (16, 76309, 104597, 25675, 53279, 132201), 1: (16, 76309, 117489, 13975, 34925, 138438),
         2: (0, 65536, 91881, 22553, 46802, 116130), 3: (0, 65536, 103206, 12276, 30679, 121609)}
    for csc, (yo, cy, crv, cgu, cgv, cbu) in k.items():
        for y in range(0, 256, 17):
            for u in range(0, 256, 15):
                for v in range(0, 256, 15):
                    c = cy * (y - yo) + 32768
                    exp = tuple(min(max(x >> 16, 0), 255) for x in
                                (c + crv * (v - 128), c - cgu * (u - 128) - cgv * (v - 128), c + cbu * (u - 128)))
                    assert O.yuv2rgb_int(csc, y, u, v) == exp


def test_lanczos_table_properties():
    for (i, o) in ((1920, 1920), (3840, 1920), (100, 37), (37, 100)):
        taps, first, w = O.lanczos_table(i, o)
        assert taps == 2 * int(np.ceil(3 * max(i / o, 1.0)))
        assert np.allclose(w.sum(axis=1), 1.0, atol=1e-6)
        assert np.all(np.diff(first) >= 0)
    # identity size: the centre tap carries all the weight
    taps, first, w = O.lanczos_table(64, 64)
    assert np.all(w.max(axis=1) > 0.9999)


def test_lanczos_constant_image_stays_constant():
    src = np.full((40, 60, 4), 173, dtype=np.uint8)
    dst = np.zeros((20, 30, 4), dtype=np.uint8)
    assert O.lanczos_bgra(dst, src) == 0
    assert np.all(dst == 173)


def test_code_scale_sampler_vs_unit_scale():
    """The BGRA-target family (spec owned by this repo) samples on the 0..255 code scale with fused
    multiply-adds; the reference's kernels sample on the unit scale through the Khronos LINEAR filter
    (c/255 per tap, sequential roundings, *255 at the store).  north_star allows the float scaler
    1 ULP.  Tolerance of this test: the two agree to within ONE CODE of the sampled value everywhere,
    and they differ only where the exact filter value lies within 1e-3 of a rounding tie (x.5) — there
    the unit-scale result is decided by its own rounding noise, the code-scale one by ties-to-even."""
    W, H, w, h = 192, 108, 128, 72
    f32 = np.float32
    u = util.full_canvas_uniforms((w, h), (W, H))

    def axis(n_out, n_in):
        o = np.arange(n_out, dtype=f32) / f32(n_out)             # out_uv = gid / size
        t = (o * f32(2) - f32(1)) * f32(.5) + f32(.5)             # transform row (.5, 0, 0, .5); textureTx = I
        um = t * f32(n_in) - f32(.5)
        fl = np.floor(um)
        i0 = fl.astype(np.int32)
        return np.clip(i0, 0, n_in - 1), np.clip(i0 + 1, 0, n_in - 1), (um - fl).astype(f32)

    x0, x1, a = axis(w, W)
    y0, y1, b = axis(h, H)
    ia, ib = f32(1) - a, f32(1) - b
    w00, w10 = ib[:, None] * ia[None, :], ib[:, None] * a[None, :]
    w01, w11 = b[:, None] * ia[None, :], b[:, None] * a[None, :]

    def taps(p):
        return p[y0][:, x0], p[y0][:, x1], p[y1][:, x0], p[y1][:, x1]

    def exact(p):       # the filter value in float64 (weights as the float32 values both paths use)
        t00, t10, t01, t11 = (t.astype(np.float64) for t in taps(p))
        return w00.astype(np.float64) * t00 + w10.astype(np.float64) * t10 + w01.astype(np.float64) * t01 + w11.astype(np.float64) * t11

    def check(own, unit, p):
        d = own.astype(np.int32) - unit.astype(np.int32)
        assert np.abs(d).max() <= 1
        e = exact(p)
        tie_distance = np.abs(np.abs(e - np.floor(e)) - 0.5)
        assert np.all(tie_distance[d != 0] < 1e-3)
        assert np.all(np.abs(own.astype(np.float64) - e) <= 0.5 + 1e-3)   # and the owned result is a correct rounding

    # (1) luma: img_nv12_bgra (full-range matrix, flat chroma => B = G = R = luma code) against the luma plane
    # the reference kernel img_nv12_nv12 writes for the same picture and uniforms (checked against the
    # compiled reference OpenCL source by test_oracle_equals_compiled_reference_kernels)
    src = util.alloc_image("nv12", W, H, seed=77)
    src[1][...] = 128
    ref = util.alloc_image("nv12", w, h)
    assert O.run_kernel("img_clear_nv12", ref) == 0
    assert O.run_kernel("img_nv12_nv12", ref, src, u) == 0
    own = util.alloc_image("bgra", w, h)
    assert O.run_kernel("img_clear_bgra", own) == 0
    assert O.run_kernel("img_nv12_bgra", own, src, u, csc=2) == 0
    assert np.array_equal(own[0][..., 0], own[0][..., 1]) and np.array_equal(own[0][..., 0], own[0][..., 2])
    check(own[0][..., 0], ref[0], src[0])
    # (2) four-channel texels: img_bgra_bgra_tx of an opaque picture (alpha 255, opacity 1, cleared canvas)
    # against a float32 numpy evaluation of the unit-scale filter
    pic = util.alloc_image("bgra", W, H, seed=78)
    pic[0][..., 3] = 255
    own = util.alloc_image("bgra", w, h)
    assert O.run_kernel("img_clear_bgra", own) == 0
    assert O.run_kernel("img_bgra_bgra_tx", own, pic, u) == 0
    for c in range(3):
        t00, t10, t01, t11 = (t.astype(f32) / f32(255) for t in taps(pic[0][..., c]))
        q = ((w00 * t00 + w10 * t10) + w01 * t01) + w11 * t11
        unit = np.clip(np.rint(q * f32(255)), 0, 255)
        check(own[0][..., c], unit, pic[0][..., c])
    assert np.all(own[0][..., 3] == 255)


def test_reference_quirks_are_preserved():
    """Bit-level quirks of the reference called out in SURVEY section 2.2."""
    # same-size full-canvas composite is a half-pixel box filter, not a copy (out_uv = gid/size, no +0.5)
    src = util.alloc_image("nv12", 16, 8, seed=1)
    dst = util.alloc_image("nv12", 16, 8)
    assert O.run_kernel("img_clear_nv12", dst) == 0
    assert O.run_kernel("img_nv12_nv12", dst, src, util.full_canvas_uniforms((16, 8), (16, 8))) == 0
    assert dst[0][0, 0] == src[0][0, 0]                      # clamped corner
    assert not np.array_equal(dst[0], src[0])
    y = src[0].astype(np.float64)
    box = (y[2, 4] + y[2, 5] + y[3, 4] + y[3, 5]) / 4
    assert abs(int(dst[0][3, 5]) - box) <= 1
    # clear: Y = 0, chroma = 0.5 -> 128 (round half to even)
    assert np.all(dst[1] != 0)
    c = util.alloc_image("y420p", 8, 4, seed=2)
    assert O.run_kernel("img_clear_y420p", c) == 0
    assert np.all(c[0] == 0) and np.all(c[1] == 128) and np.all(c[2] == 128)
    # pure blue through img_bgra_nv12: Y = 0.113 * 255 -> 29 (the reference's 0.113, not 0.114)
    blue = np.zeros((8, 16, 4), dtype=np.uint8)
    blue[..., 0] = 255
    blue[..., 3] = 255
    nv = util.alloc_image("nv12", 16, 8)
    assert O.run_kernel("img_clear_nv12", nv) == 0
    assert O.run_kernel("img_bgra_nv12", nv, [blue], util.full_canvas_uniforms((16, 8), (16, 8))) == 0
    assert np.all(nv[0] == 29) and np.all(nv[1][..., 0] == 255) and np.all(nv[1][..., 1] == 107)
    # RGB-source kernels write nothing outside tx in [0,1]^2, YUV-source kernels paint the border
    u = util.make_uniforms((32, 16), rect=(8, 4, 16, 8), border=(4, 2, 4, 2), fill=(1, 0, 0, 1), in_size=(16, 8))
    c0 = util.alloc_image("nv12", 32, 16, seed=3)
    a, b = util.copy_image(c0), util.copy_image(c0)
    assert O.run_kernel("img_bgra_nv12", a, util.alloc_image("bgra", 16, 8, seed=4), u) == 0
    assert O.run_kernel("img_nv12_nv12", b, util.alloc_image("nv12", 16, 8, seed=4), u) == 0
    assert np.array_equal(a[0][2:4, 4:28], c0[0][2:4, 4:28])          # border rows untouched by the RGB kernel
    assert not np.array_equal(b[0][2:4, 4:28], c0[0][2:4, 4:28])       # painted by the YUV kernel
    assert np.array_equal(b[0][:2], c0[0][:2])                          # outside the border quad: untouched


def test_oracle_argument_checks():
    nv = util.alloc_image("nv12", 16, 8)
    bg = util.alloc_image("bgra", 16, 8)
    u = util.full_canvas_uniforms((16, 8), (16, 8))
    assert O.run_kernel("img_clear_yuvs", nv) == 6          # enum case without a kernel
    assert O.run_kernel("img_bgra_nv12", bg, bg, u) == 4    # bad target
    assert O.run_kernel("img_bgra_nv12", nv, nv, u) == 5    # bad input
    assert O.run_kernel("img_bgra_nv12", nv, bg, None) == 1


# ---- the whole owned family against a unit-scale evaluation in the reference family's style -----------------------
# oracle/ref_kernels.c::px_to_bgra_unit evaluates img_{nv12,y420p}_bgra / img_{bgra,rgba}_bgra_tx the way the reference's
# OpenCL kernels evaluate theirs: texels through c/255, the Khronos filter with sequential roundings, unfused blends in
# the source order of px_yuv_to_yuv / px_rgb_to_yuv, *255 RTE at the store.  The code-scale specification
# is held to that evaluation here, class by class.  Bounds (codes, per output channel):
#   RGB sources: <= 1 in EVERY class, stacks of 2..8 re-quantised layers included — the two evaluations round the same exact
#                value and part only at rounding ties; a one-code difference does not grow through later layers because it
#                is scaled by (1 - alpha) <= 1 and re-rounded.
#   YUV sources: the two samplers differ by <= 1 code in Y, U or V (at ties), and the INTEGER colour matrix that follows —
#                identical in both — amplifies that: |dB| <= (cy + cbu) / 65536 = 3.3, |dG| <= 2.4, |dR| <= 3.0, so <= 4.  That is a
#                property of putting an integer matrix behind ANY float sampler that is allowed 1 ULP, not of the code scale.
ENVELOPE_BOUND = {"rgb": 1, "yuv": 4}
ENVELOPE_OBSERVED = {   # max |code-scale - unit-scale| seen per class (asserted as upper bounds; they document the envelope)
    ("scale", "rgb"): 1, ("alpha", "rgb"): 1, ("fill", "rgb"): 1, ("rotated", "rgb"): 1, ("stack", "rgb"): 1,
    ("scale", "yuv"): 3, ("alpha", "yuv"): 3, ("fill", "yuv"): 3, ("rotated", "yuv"): 3, ("stack", "yuv"): 3,
}
_ENV_KINDS = {"yuv": ["img_nv12_bgra", "img_y420p_bgra"], "rgb": ["img_bgra_bgra_tx", "img_rgba_bgra_tx"]}


def _envelope_layer(rng, cw, ch, cls, kinds):
    k = kinds[int(rng.integers(0, len(kinds)))]
    s = k.split("_")[1]
    sw, sh = int(rng.integers(8, 120)) * 2, int(rng.integers(4, 70)) * 2
    kw = {}
    if cls != "scale" and rng.random() < 0.7:
        kw["rect"] = (float(rng.uniform(-0.3, 0.6) * cw), float(rng.uniform(-0.3, 0.6) * ch),
                      float(rng.uniform(0.2, 1.5) * cw), float(rng.uniform(0.2, 1.5) * ch))
    if cls in ("fill", "rotated", "stack"):
        if rng.random() < 0.6:
            kw["border"] = tuple(float(v) for v in rng.uniform(0, 10, 4))
        if rng.random() < 0.6:
            kw["fill"] = tuple(float(v) for v in rng.uniform(0, 1, 4))
    if cls == "rotated" or (cls == "stack" and rng.random() < 0.3):
        kw["rotation"] = float(rng.uniform(-0.6, 0.6))
    if cls != "scale":
        kw["opacity"] = float(rng.choice([1.0, rng.uniform(0, 1)]))
    if rng.random() < 0.4:
        kw["tex"] = (float(rng.uniform(0, 0.4)), float(rng.uniform(0, 0.4)),
                     float(rng.uniform(0.3, 1.0)) * (1 if rng.random() < 0.8 else -1), float(rng.uniform(0.3, 1.0)))
    u = util.make_uniforms((cw, ch), in_size=(sw, sh), **kw)
    src = util.alloc_image(s, sw, sh, seed=int(rng.integers(1, 1 << 20)))
    if cls == "scale" and s in ("bgra", "rgba"):
        src[0][..., 3] = 255                                   # opaque pictures: the sampler alone
    return k, src, u, int(rng.integers(0, 4))


@pytest.mark.parametrize("family", ["rgb", "yuv"])
@pytest.mark.parametrize("cls", ["scale", "alpha", "fill", "rotated", "stack"])
def test_code_scale_family_envelope(cls, family):
    """scale: opaque full-canvas pictures at random up / down scales (the sampler alone);  alpha: per-pixel alpha and
    opacity < 1, partial cover;  fill: borders and fill colours with alpha;  rotated: rotated quads;  stack: 2..8 layers
    with the canvas re-quantised in between, un-cleared canvases included."""
    worst, differing, total, touched = 0, 0, 0, 0
    for t in range(36):
        rng = np.random.default_rng(zlib.crc32(f"{cls}/{family}".encode()) % 1000 * 1000 + t)
        cw, ch = int(rng.integers(16, 200)), int(rng.integers(8, 120))
        c0 = util.alloc_image("bgra", cw, ch, seed=int(rng.integers(1, 1 << 20)))
        own, unit = util.copy_image(c0), util.copy_image(c0)
        if rng.random() < 0.5:
            assert O.run_kernel("img_clear_bgra", own) == 0 and O.run_kernel("img_clear_bgra", unit) == 0
        before = own[0].copy()
        for _ in range(int(rng.integers(2, 9)) if cls == "stack" else 1):
            k, src, u, csc = _envelope_layer(rng, cw, ch, cls, _ENV_KINDS[family])
            assert O.run_kernel(k, own, src, u, csc=csc) == 0
            assert O.run_kernel(O.ENVELOPE_IDS[k], unit, src, u, csc=csc) == 0
        d = np.abs(own[0][..., :3].astype(int) - unit[0][..., :3].astype(int))
        worst = max(worst, int(d.max()))
        differing += int((d != 0).sum()); total += d.size
        touched += int((own[0] != before).any(axis=2).sum())
        assert np.array_equal(own[0][..., 3], unit[0][..., 3])              # alpha bytes: identical rules
    assert touched > 0.1 * total / 3, "the cases must actually paint"
    assert worst <= ENVELOPE_BOUND[family], (cls, family, worst)
    assert worst <= ENVELOPE_OBSERVED[(cls, family)], f"{cls}/{family}: deviation {worst} exceeds the recorded envelope"
    assert differing <= 2e-3 * total, f"{cls}/{family}: {differing} of {total} channel values differ — more than rounding ties explain"
