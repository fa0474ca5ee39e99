#!/usr/bin/env python3
"""tools/csc_absorb_search.py — biases of the float -> code conversion that absorb the red and blue channel offsets of the integer YUV -> RGB
matrix (pixel_math.hip.h, kCscAbsorb).

The kernels turn a filtered sample f (a float in code scale, 0..255) into an integer operand of v_mad_i32_i24 with one float add:
bits(f + M) for M = 2^23 + B is 0x4B000000 + B + rint(f), and the 24-bit multiplier reads B + rint(f) (B even: ties keep going to the even
code; B + 255 < 2^23: bit 23 stays clear).  With M = -(2^23 + B) the operand is B - rint(f) and the layer's coefficients for that operand are
negated.  A bias B on an operand adds coefficient x B to every channel the operand feeds, in wrap-around 32-bit arithmetic:

    r = crv (V + bv) + cy (Y + by)                      = R - KR + (crv bv + cy by)
    b = cbu (U + bu) + cy (Y + by)                      = Bl - KB + (cbu bu + cy by)
    g = -cgu (U + bu) - cgv (V + bv) + cy (Y + by) + kg'

so biases with  crv bv + cy by = KR  and  cbu bu + cy by = KB  (mod 2^32) make the red and the blue channel come out of their last
multiply-add finished — two vector adds fewer per pixel and layer — and kg' = KG - (cy by - cgu bu - cgv bv) keeps green right.
Prints the table for the four matrices of kCsc (None: no such biases exist — BT.601 full range)."""
import numpy as np

M32 = 1 << 32
LIM = (1 << 23) - 256
MATRICES = [("BT.601 limited", (16, 76309, 104597, 25675, 53279, 132201)), ("BT.709 limited", (16, 76309, 117489, 13975, 34925, 138438)),
            ("BT.601 full", (0, 65536, 91881, 22553, 46802, 116130)), ("BT.709 full", (0, 65536, 103206, 12276, 30679, 121609))]


def constants(m):
    yoff, cy, crv, cgu, cgv, cbu = m
    base = 32768 - cy * yoff
    return (base - 128 * crv) % M32, (base + 128 * (cgu + cgv)) % M32, (base - 128 * cbu) % M32


def solve(c, need):
    """even signed bx in (-LIM, LIM) with c * bx == need (mod 2^32) per element; 2^40 where there is none"""
    k = (c & -c).bit_length() - 1
    mod = 1 << (32 - k)
    ok = (need & np.uint64((1 << k) - 1)) == 0
    bx = (((need >> np.uint64(k)) * np.uint64(pow(c >> k, -1, mod))) % np.uint64(mod)).astype(np.int64)
    pos = ok & (bx < LIM) & ((bx & 1) == 0)
    neg = ok & ((mod - bx) < LIM) & (((mod - bx) & 1) == 0)
    return np.where(pos, bx, np.where(neg, bx - mod, np.int64(1 << 40)))


def search(m):
    yoff, cy, crv, cgu, cgv, cbu = m
    KR, KG, KB = constants(m)
    by = np.arange(256, LIM, 2, dtype=np.int64)        # (a negative bias B - code needs B >= 255: the same bound on both signs keeps it simple)
    by = np.concatenate([by, -by])
    t = (by * cy) % M32
    bv = solve(crv, ((KR - t) % M32).astype(np.uint64))
    bu = solve(cbu, ((KB - t) % M32).astype(np.uint64))
    ok = (np.abs(bv) < LIM) & (np.abs(bu) < LIM) & (np.abs(bv) >= 256) & (np.abs(bu) >= 256)
    idx = np.nonzero(ok)[0]
    if len(idx) == 0:
        return None
    # all-positive solutions first, then the smallest luma bias
    best = min(idx, key=lambda i: ((by[i] < 0) + (bu[i] < 0) + (bv[i] < 0), abs(int(by[i]))))
    BY, BU, BV = int(by[best]), int(bu[best]), int(bv[best])
    assert (cy * BY + crv * BV - KR) % M32 == 0 and (cy * BY + cbu * BU - KB) % M32 == 0
    kg = (KG - (cy * BY - cgu * BU - cgv * BV)) % M32
    return BY, BU, BV, kg, len(idx)


def table():
    return [(name, search(m)) for name, m in MATRICES]


def folded_rows():
    """the rows of kCscAbsorbed (pixel_math.hip.h): coefficients with their operand's sign, kg' as int32, the conversion constants"""
    rows = []
    for name, m in MATRICES:
        s = search(m)
        yoff, cy, crv, cgu, cgv, cbu = m
        if s is None:
            rows.append((name, None)); continue
        BY, BU, BV, kg, n = s
        sg = lambda b: -1 if b < 0 else 1
        sy, su, sv = sg(BY), sg(BU), sg(BV)
        rows.append((name, (sy * cy, sv * crv, -su * cgu, -sv * cgv, su * cbu, kg - (1 << 32) if kg >= (1 << 31) else kg,
                            float(sy * (8388608 + abs(BY))), float(su * (8388608 + abs(BU))), float(sv * (8388608 + abs(BV))), (BY, BU, BV))))
    return rows


if __name__ == "__main__":
    for name, r in folded_rows():
        if r is None:
            print(f"    {{ 0, 0, 0, 0, 0, 0, 0.f, 0.f, 0.f }},   // {name}: no such biases")
        else:
            print(f"    {{ {r[0]}, {r[1]}, {r[2]}, {r[3]}, {r[4]}, {r[5]}, {r[6]:.1f}f, {r[7]:.1f}f, {r[8]:.1f}f }},   // {name}: biases {r[9][0]}, {r[9][1]}, {r[9][2]}")
