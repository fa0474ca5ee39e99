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
