"""The absorbed form of the integer YUV -> RGB matrix (swiftvideo_amd/csrc/pixel_math.hip.h, kCscAbsorbed): conversion biases that leave the
red and the blue channel finished after their last multiply-add.  CPU side: the table in the header is the one tools/csc_absorb_search.py
derives; it satisfies the bounds the float adder needs; and — in numpy, with the multiplier's 24-bit operands and 32-bit wrap-around spelled
out — it gives the plain formula's three 16.16 sums for ALL 2^24 code triples of each matrix.  (Device side: tests/test_gpu_matrices.py.)"""
import re
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tools"))
import csc_absorb_search as S  # noqa: E402

from test_first_principles import CSC  # noqa: E402  (the four matrices, typed in from the standards' Kr / Kb)


def header_rows():
    text = (ROOT / "swiftvideo_amd" / "csrc" / "pixel_math.hip.h").read_text()
    body = text[text.index("kCscAbsorbed[4] = {"):]
    body = body[: body.index("};")]
    rows = []
    for m in re.finditer(r"\{\s*(-?\d+),\s*(-?\d+),\s*(-?\d+),\s*(-?\d+),\s*(-?\d+),\s*(-?\d+),\s*(-?[\d.]+)f,\s*(-?[\d.]+)f,\s*(-?[\d.]+)f\s*\}", body):
        g = m.groups()
        rows.append(tuple(int(x) for x in g[:6]) + tuple(float(x) for x in g[6:]))
    assert len(rows) == 4
    return rows


def test_tool_matrices_are_the_standards():
    assert [tuple(m) for _, m in S.MATRICES] == [tuple(CSC[i]) for i in range(4)]


def test_header_table_is_what_the_search_derives():
    for got, (name, want) in zip(header_rows(), S.folded_rows()):
        if want is None:
            assert got == (0, 0, 0, 0, 0, 0, 0.0, 0.0, 0.0), name
        else:
            assert got == tuple(want[:9]), name
    # BT.601 full range is the one without such biases (csc_absorbable in the header says so)
    assert [r is None for _, r in S.folded_rows()] == [False, False, True, False]
    assert "constexpr bool csc_absorbable(int csc) { return (csc & 3) != 2; }" in (ROOT / "swiftvideo_amd" / "csrc" / "pixel_math.hip.h").read_text()


@pytest.mark.parametrize("csc", [0, 1, 3])
def test_conversion_constants_keep_the_float_adders_contract(csc):
    row = header_rows()[csc]
    for mag in row[6:]:
        b = abs(mag) - (1 << 23)
        assert mag == float(np.float32(mag)) and b == int(b)            # exactly a float
        assert b % 2 == 0, "an odd bias would send ties to the odd code"
        assert 256 <= b < (1 << 23) - 256, "bit 23 must stay clear over the 256 codes (and B - code positive for the negative form)"


@pytest.mark.parametrize("csc", [0, 1, 3])
def test_absorbed_form_gives_the_plain_sums_for_every_triple(csc):
    cy, crv, ncgu, ncgv, cbu, kg, my, mu, mv = header_rows()[csc]
    yoff, pcy, pcrv, pcgu, pcgv, pcbu = CSC[csc]
    M32 = 1 << 32

    def operand(code, mag):
        # bits(code + M) & 0xFFFFFF, read as a signed 24-bit number by v_mul_i32_i24 / v_mad_i32_i24
        b = int(abs(mag)) - (1 << 23)
        low = (b + code) if mag > 0 else (b - code)
        assert low.min() >= 0 and low.max() < (1 << 23)
        return low

    u, v = np.meshgrid(np.arange(256, dtype=np.int64), np.arange(256, dtype=np.int64), indexing="ij")
    uo, vo = operand(u, mu), operand(v, mv)
    wrap = lambda t: ((t + (1 << 31)) % M32) - (1 << 31)        # noqa: E731  (int32 wrap-around)
    for y in range(256):
        yo = operand(np.int64(y) + np.zeros((), np.int64), my)
        t = yo * cy
        r = wrap(vo * crv + t)
        g = wrap(vo * ncgv + uo * ncgu + t + kg)
        b = wrap(uo * cbu + t)
        c = pcy * (y - yoff) + 32768
        d, e = u - 128, v - 128
        assert np.array_equal(r, c + pcrv * e) and np.array_equal(g, c - pcgu * d - pcgv * e) and np.array_equal(b, c + pcbu * d), (csc, y)
