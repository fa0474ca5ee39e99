"""Device primitives against the oracle's definitions, exhaustively where the domain is small."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O
from swiftvideo_amd import chipvideo as cv

pytestmark = pytest.mark.gpu


def _selftest(ctx, in_f, num, den):
    lib = cv.load()
    fn = lib.chv_selftest_primitives
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p] + [C.c_void_p] * 6 + [C.c_int]
    n = in_f.size
    un = np.zeros(256, dtype=np.float32)
    codes = np.zeros(n, dtype=np.uint8)
    quot = np.zeros(n, dtype=np.float32)
    cv.check(fn(ctx.handle, un.ctypes.data, in_f.ctypes.data, codes.ctypes.data, num.ctypes.data,
                den.ctypes.data, quot.ctypes.data, n))
    return un, codes, quot


def test_unorm8_store_and_division_match_oracle(ctx):
    rng = np.random.default_rng(1)
    # values around every rounding boundary k+0.5 (in code units), plus specials
    k = np.arange(-2, 259, dtype=np.float64)
    edge = np.concatenate([(k + 0.5 + d) / 255.0 for d in (-1e-4, -1e-6, 0.0, 1e-6, 1e-4)])
    vals = np.concatenate([edge, rng.uniform(-0.5, 1.5, 20000), [np.nan, np.inf, -np.inf, 0.0, -0.0, 1.0]])
    in_f = vals.astype(np.float32)
    n = in_f.size
    # the divisions `geometry` performs: gid / size for the BASELINE canvas sizes and odd ones
    sizes = np.array([1280, 720, 1920, 1080, 3840, 2160, 7, 5, 33, 17, 64, 36], dtype=np.float32)
    den = rng.choice(sizes, n).astype(np.float32)
    num = np.floor(rng.uniform(0, 1, n).astype(np.float32) * den).astype(np.float32)
    un, codes, quot = _selftest(ctx, in_f, num, den)
    lib = O.lib()
    exp_un = np.array([lib.orc_load_unorm8(c) for c in range(256)], dtype=np.float32)
    assert np.array_equal(un.view(np.uint32), exp_un.view(np.uint32)), "unorm8(c) != c/255.0f"
    exp_codes = np.array([lib.orc_store_unorm8(float(f)) for f in in_f], dtype=np.uint8)
    bad = np.nonzero(codes != exp_codes)[0]
    assert bad.size == 0, f"to_code differs for {in_f[bad[:5]]}: {codes[bad[:5]]} vs {exp_codes[bad[:5]]}"
    assert np.array_equal(quot.view(np.uint32), (num / den).astype(np.float32).view(np.uint32)), \
        "device float division is not correctly rounded"


def test_packed_saturating_shift_matches_plain_packing(ctx):
    """pack_bgra_fixed_pk (v_ashr_pk_u8_i32 + v_perm_b32) == pack_bgra_fixed (clamp, shift, or) for 16.16 sums
    over the whole range the colour matrices can produce, and == numpy."""
    lib = cv.load()
    fn = lib.chv_selftest_pack
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p] + [C.c_void_p] * 4 + [C.c_int]
    rng = np.random.default_rng(3)
    n = 200000
    chans = [np.concatenate([rng.integers(-40_000_000, 60_000_000, n - 12),
                             [0, 65535, 65536, -1, -65536, 255 * 65536, 255 * 65536 + 65535, 256 * 65536, 2**31 - 1, -2**31, 16777215, 16777216]]).astype(np.int32)
             for _ in range(3)]
    for c in chans:
        rng.shuffle(c)
    out = np.zeros(2 * n, dtype=np.uint32)
    cv.check(fn(ctx.handle, chans[0].ctypes.data, chans[1].ctypes.data, chans[2].ctypes.data, out.ctypes.data, n))
    clip = lambda v: np.clip(v.astype(np.int64) >> 16, 0, 255).astype(np.uint32)
    exp = clip(chans[0]) | (clip(chans[1]) << 8) | (clip(chans[2]) << 16) | np.uint32(0xFF000000)
    assert np.array_equal(out[0::2], exp), "pack_bgra_fixed"
    bad = np.nonzero(out[1::2] != exp)[0]
    assert bad.size == 0, f"pack_bgra_fixed_pk differs at {bad[:5]}: {[hex(v) for v in out[1::2][bad[:5]]]} vs {[hex(v) for v in exp[bad[:5]]]}"


def test_pack_codes_matches_round_clamp_pack(ctx):
    """pack_codes (v_cvt_pk_u8_f32 per byte) == clamp(rint(x), 0, 255) per channel with NaN -> 0, packed: ties, values around
    both clamps, huge magnitudes, infinities, NaN, denormals."""
    lib = cv.load()
    fn = lib.chv_selftest_pack_codes
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    rng = np.random.default_rng(5)
    k = np.arange(-3, 260, dtype=np.float64)
    edge = np.concatenate([k + 0.5 + d for d in (-1e-3, -2e-5, 0.0, 2e-5, 1e-3)] + [k])
    special = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 1e-45, -1e-45, 1e30, -1e30, 255.49998, 255.5, 255.50002, 0.49999997, 0.5], dtype=np.float64)
    vals = np.concatenate([edge, special, rng.uniform(-40, 300, 40000)]).astype(np.float32)
    vals = np.resize(vals, (vals.size + 3) // 4 * 4)
    rng.shuffle(vals)
    n = vals.size // 4
    out = np.zeros(2 * n, dtype=np.uint32)
    cv.check(fn(ctx.handle, vals.ctypes.data, out.ctypes.data, n))
    with np.errstate(invalid="ignore"):
        code = np.where(np.isnan(vals), 0.0, np.clip(np.rint(vals.astype(np.float64)), 0, 255)).astype(np.uint32).reshape(n, 4)
    exp = code[:, 0] | (code[:, 1] << 8) | (code[:, 2] << 16) | (code[:, 3] << 24)
    assert np.array_equal(out[0::2], exp), "to_code_raw"
    bad = np.nonzero(out[1::2] != exp)[0]
    assert bad.size == 0, f"pack_codes differs for {vals.reshape(n, 4)[bad[:3]]}: {[hex(v) for v in out[1::2][bad[:3]]]} vs {[hex(v) for v in exp[bad[:3]]]}"
