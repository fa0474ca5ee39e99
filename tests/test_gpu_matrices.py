"""The one thing north_star calls bit-exact — the integer BT.601 / BT.709 matrices, both directions — proven EXHAUSTIVELY on the device:
all 2^24 (Y, U, V) triples through the YUV -> RGB matrix and all 2^24 (R, G, B) triples through the RGB -> YUV matrix, for each of the four
colourspaces, against the integer formulas of DESIGN.md 4.2 / 4.5 evaluated here in numpy (coefficient tables typed in / rebuilt from Kr, Kb in
rational arithmetic by tests/test_first_principles.py and tests/test_rgb_to_yuv_int.py — not through oracle/ref_kernels.c).  The device kernel
(chv_selftest_matrices) also runs every OTHER form of the YUV -> RGB matrix the production kernels use — folded offsets, operands carrying the
float adder's bias packed by v_ashr_pk_u8_i32, float codes packed by v_cvt_pk_u8_f32 — and counts the triples on which any of them differs."""
import ctypes as C

import numpy as np
import pytest

from test_first_principles import CSC
from test_rgb_to_yuv_int import tables

pytestmark = pytest.mark.gpu


def _device(ctx, direction, csc):
    from swiftvideo_amd import chipvideo as cv
    lib = cv.load()
    fn = lib.chv_selftest_matrices
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    out = np.empty(1 << 24, dtype=np.uint32)
    mism = C.c_uint32(0xFFFFFFFF)
    cv.check(fn(ctx.handle, direction, csc, out.ctypes.data, C.byref(mism)))
    return out, mism.value


def _triples():
    i = np.arange(1 << 24, dtype=np.int64)
    return i >> 16, (i >> 8) & 255, i & 255


@pytest.mark.parametrize("csc", [0, 1, 2, 3])
def test_yuv_to_rgb_every_triple(ctx, csc):
    got, mism = _device(ctx, 0, csc)
    y, u, v = _triples()
    yoff, cy, crv, cgu, cgv, cbu = CSC[csc]
    c = cy * (y - yoff) + 32768
    d, e = u - 128, v - 128
    clip = lambda t: np.clip(t >> 16, 0, 255)      # noqa: E731  (>> on int64 is arithmetic)
    b, g, r = clip(c + cbu * d), clip(c - cgu * d - cgv * e), clip(c + crv * e)
    exp = (b | (g << 8) | (r << 16) | (255 << 24)).astype(np.uint32)
    bad = np.flatnonzero(got != exp)
    assert bad.size == 0, f"csc {csc}: {bad.size} of 2^24 triples differ, first (Y, U, V) = {(int(bad[0]) >> 16, (int(bad[0]) >> 8) & 255, int(bad[0]) & 255)}: " \
                          f"device {int(got[bad[0]]):08x}, formula {int(exp[bad[0]]):08x}"
    assert mism == 0, f"csc {csc}: the folded / biased / float-code forms of the matrix differ from the plain form on {mism} triples"


@pytest.mark.parametrize("csc", [0, 1, 2, 3])
def test_rgb_to_yuv_every_triple(ctx, csc):
    got, mism = _device(ctx, 1, csc)
    r, g, b = _triples()
    yoff, ky, ku, kv = tables(csc)
    clip = lambda t: np.clip(t >> 16, 0, 255)      # noqa: E731
    yy = clip(ky[0] * r + ky[1] * g + ky[2] * b + (yoff << 16) + 32768)
    uu = clip(ku[0] * r + ku[1] * g + ku[2] * b + (128 << 16) + 32768)
    vv = clip(kv[0] * r + kv[1] * g + kv[2] * b + (128 << 16) + 32768)
    exp = (yy | (uu << 8) | (vv << 16)).astype(np.uint32)
    bad = np.flatnonzero(got != exp)
    assert bad.size == 0, f"csc {csc}: {bad.size} of 2^24 triples differ, first (R, G, B) = {(int(bad[0]) >> 16, (int(bad[0]) >> 8) & 255, int(bad[0]) & 255)}: " \
                          f"device {int(got[bad[0]]):06x}, formula {int(exp[bad[0]]):06x}"
    assert mism == 0, f"csc {csc}: the tick kernels' form of the rows (biased operands, folded offsets, float codes) differs from rgb_to_yuv_int on {mism} triples"
