"""The two kernels of `enum ComputeKernel` no caller of the reference dispatches (compute.swift:67,70):
snd_s16i_s16i (kernels.cl.swift:534-562) and me_fullsearch (kernels.metal:129-267).

CPU part: the oracle's restatements against (a) the reference's own snd_s16i_s16i kernel string compiled for x86-64 (oracle/clref — the one
kernel whose compiled form needs nothing of an OpenCL runtime but get_global_id and min) and (b) a second, independent statement of the Metal
source in plain Python.  GPU part: the HIP kernels through chv_run_kernel against the oracle, bit for bit."""
import ctypes as C
import math
import struct

import numpy as np
import pytest

from oracle import oracle as O


# ---- snd_s16i_s16i ------------------------------------------------------------------------------------------------------------
def snd_cases():
    rng = np.random.default_rng(77)
    full = lambda n: rng.integers(-32768, 32768, n).astype(np.int16)      # noqa: E731
    yield "three_inputs", full(1001), [full(1001) for _ in range(3)], [0.7, 1.9, -2.5], [0.25, 0.5, 0.9]
    yield "eight_inputs", full(4096), [full(4096) for _ in range(8)], list(rng.uniform(-1.5, 1.5, 8)), list(rng.uniform(0, 1, 8))
    yield "none", full(64), [], [], []
    yield "odd_length_1", full(1), [full(1)], [1.0], [0.5]
    yield "odd_length_7", full(7), [full(7), full(7)], [0.5, 0.5], [0.0, 1.0]
    # saturation: the cap at +32767 (min), and below -32768 the wrap of the x86 conversion (no max in the source)
    edge = np.array([32767, -32768, 32767, -32768, 1, -1, 0, 12345] * 4, dtype=np.int16)
    yield "cap_and_wrap", np.zeros(32, dtype=np.int16), [edge, edge], [4.0, -4.0], [0.0, 1.0]
    yield "huge_gain", full(256), [full(256)], [3.0e9], [0.5]             # beyond int32: INT32_MIN -> low half 0
    yield "fade_ends", full(130), [full(130), full(130)], [1.0, 1.0], [0.0, 1.0]
    yield "accumulator_wraps", np.full(66, 32000, dtype=np.int16), [np.full(66, 30000, dtype=np.int16)], [1.0], [0.5]


@pytest.mark.parametrize("case", [c[0] for c in snd_cases()])
def test_snd_oracle_equals_the_compiled_reference_kernel(case):
    if O.clref() is None or not hasattr(O.clref(), "clref_run_snd"):
        pytest.skip("libclref.so with snd_s16i_s16i not built (needs /root/reference at build time)")
    _, out0, ins, gains, fades = next(c for c in snd_cases() if c[0] == case)
    u = O.snd_uniforms(gains, fades)
    a, b = out0.copy(), out0.copy()
    assert O.snd_s16i_s16i(a, ins, u) == 0 and O.clref_snd_s16i_s16i(b, ins, u) == 0
    assert np.array_equal(a, b)
    if ins:
        assert not np.array_equal(a, out0)


def test_snd_first_principles():
    """unity gain, centre fade: each channel gets half of the input; left-only and right-only fades route a mono source"""
    x = np.array([1000, 1000, -2000, -2000, 7, 7], dtype=np.int16)
    out = np.zeros(6, dtype=np.int16)
    assert O.snd_s16i_s16i(out, [x], O.snd_uniforms([1.0], [0.5])) == 0
    assert out.tolist() == [500, 500, -1000, -1000, 3, 3]                  # (short) truncates toward zero: 3.5 -> 3
    out[:] = 0
    assert O.snd_s16i_s16i(out, [x], O.snd_uniforms([1.0], [0.0])) == 0     # fade 0: everything left
    assert out.tolist() == [1000, 0, -2000, 0, 7, 0]
    out[:] = 0
    assert O.snd_s16i_s16i(out, [x], O.snd_uniforms([1.0], [1.0])) == 0
    assert out.tolist() == [0, 1000, 0, -2000, 0, 7]


# ---- me_fullsearch: an independent statement of kernels.metal:129-267 -----------------------------------------------------------
F = np.float32


def me_python(ref, cur, block, window, image, trace=None, honour_early_exit=True):
    bsx, bsy = block
    cost = O.me_cost_table(256)
    swx, swy = min(window[0], 64), min(window[1], 64)
    maxx, maxy = F(window[0] // 2), F(window[1] // 2)
    rd = lambda p, x, y: F(p[y, x]) / F(255.0) if 0 <= x < p.shape[1] and 0 <= y < p.shape[0] else F(0)      # noqa: E731
    clamp = lambda v, lo, hi: min(max(v, lo), hi)                                                               # noqa: E731
    nbx, nby = cur.shape[1] // bsx, cur.shape[0] // bsy
    out = np.zeros((nby, nbx, 4), dtype=np.uint8)
    for by in range(nby):
        for bx in range(nbx):
            ox, oy = bx * bsx, by * bsy
            left = clamp(ox + bsx // 2 - swx // 2, 0, image[0]); top = clamp(oy + bsy // 2 - swy // 2, 0, image[1])
            right = clamp(left + swx, 0, image[0]); bottom = clamp(top + swy, 0, image[1])
            best, bmx, bmy = F(3.402823466e+38), F(0), F(0)
            rx = left
            early_exit = False                                  # kernels.metal:228
            while rx + bsx < right and not early_exit:
                side, prev = F(0), F(0)
                ry = top
                while ry + bsy < bottom and not early_exit:
                    tp, sm = F(0), F(0)
                    if side > 0:
                        for x in range(bsx):
                            tp = F(tp + abs(F(rd(cur, ox + x, oy) - rd(ref, rx + x, ry))))
                        sm = F(F(prev - side) + F(0))
                    else:
                        for y in range(bsy):
                            for x in range(bsx):
                                d = abs(F(rd(cur, ox + x, oy + y) - rd(ref, rx + x, ry + y)))
                                sm = F(sm + d)
                                if y == 0:
                                    tp = F(tp + d)
                    mx, my = ox - rx, oy - ry
                    score = F(F(F(4.0) * F(cost[abs(mx)] + cost[abs(my)])) + F(sm * F(256.0)))
                    prev, side = sm, tp
                    if score < best:
                        best, bmx, bmy = score, clamp(F(mx), -maxx, maxx), clamp(F(my), -maxy, maxy)
                    if score < F(0) and honour_early_exit:      # threshold = 0, :221,247-249
                        early_exit = True
                        if trace is not None:
                            trace.append((bx, by, rx - left, ry - top, float(score)))
                    ry += 1
                rx += 1
            with np.errstate(divide="ignore", invalid="ignore"):
                vx, vy = F(F(bmx / maxx) * F(0.5) + F(0.5)), F(F(bmy / maxy) * F(0.5) + F(0.5))
            out[by, bx] = [O.lib().orc_store_unorm8(float(vx)), 128, O.lib().orc_store_unorm8(float(vy)), 255]
    return out


def me_frames(seed, w, h, smooth=False):
    rng = np.random.default_rng(seed)
    big = rng.integers(0, 256, (h + 32, w + 32)).astype(np.uint8)
    if smooth:
        yy, xx = np.mgrid[0:h + 32, 0:w + 32]
        big = ((np.sin(xx / 7.0) + np.cos(yy / 5.0)) * 60 + 128 + rng.integers(-3, 4, big.shape)).clip(0, 255).astype(np.uint8)
    dx, dy = int(rng.integers(-6, 7)), int(rng.integers(-6, 7))
    return np.ascontiguousarray(big[16:16 + h, 16:16 + w]), np.ascontiguousarray(big[16 + dy:16 + dy + h, 16 + dx:16 + dx + w])


ME_CASES = [  # (seed, w, h, block, window, smooth)
    (1, 64, 48, (16, 16), (32, 32), False), (2, 48, 40, (8, 8), (24, 16), True), (3, 40, 40, (8, 4), (64, 64), False),
    (4, 36, 30, (6, 5), (13, 11), True), (5, 32, 32, (16, 16), (200, 200), False), (6, 32, 32, (8, 8), (1, 1), False),
]


def test_cost_table_is_deltaCost2():
    t = O.me_cost_table(70)
    for d in (0, 1, 2, 7, 31, 64):
        want = F(F(4.0) * F(F(F(F(math.log2(d + 1)) * F(2.0)) + F(0.718)) + F(1.0 if d else 0.0))) + F(0.5)
        assert abs(float(t[d]) - float(want)) <= 2e-6 * max(1.0, float(want))       # (the host's log2f, not Python's: last-bit agreement is not required here)
    assert t[0] == F(F(4.0) * F(0.718)) + F(0.5)


@pytest.mark.parametrize("case", range(len(ME_CASES)))
def test_me_oracle_equals_the_python_statement(case):
    seed, w, h, block, window, smooth = ME_CASES[case]
    ref, cur = me_frames(seed, w, h, smooth)
    exp = me_python(ref, cur, block, window, (w, h))
    out = np.zeros_like(exp)
    assert O.me_fullsearch(out, ref, cur, block, window) == 0
    assert np.array_equal(out, exp)
    assert (out[:, :, 1] == 128).all() and (out[:, :, 3] == 255).all()


def test_me_sliding_window_is_the_reference_s_not_a_clean_search():
    """kernels.metal:152-166: after a candidate whose top-row SAD is > 0 the source returns previousSad - previousSide (its second loop never
    runs).  A picture shifted by a known vector is therefore NOT generally found — what is restated is the source, not the intention."""
    ref, _ = me_frames(9, 96, 64)
    cur = np.roll(ref, (2, 3), axis=(0, 1))
    out = np.zeros((4, 6, 4), dtype=np.uint8)
    assert O.me_fullsearch(out, ref, cur, (16, 16), (32, 32)) == 0
    clean = O.lib().orc_store_unorm8(float(F(F(3.0) / F(16.0)) * F(0.5) + F(0.5)))
    assert not (out[1:-1, 1:-1, 0] == clean).all()


def test_me_early_exit_is_taken_in_tall_windows_and_decides_the_vector():
    """kernels.metal:221,228-232,247-249: `threshold = 0` looks like a dead early exit, but the sliding form's running value (previousSad -
    previousSide, row after row) falls below zero once a column of candidates is a little taller than the block — the first negative score in
    visiting order ends the search and IS the answer.  ME_CASES[2] (8 x 4 blocks, 64 x 64 window) takes it in every interior block."""
    seed, w, h, block, window, smooth = ME_CASES[2]
    ref, cur = me_frames(seed, w, h, smooth)
    trace = []
    exp = me_python(ref, cur, block, window, (w, h), trace=trace)
    assert len(trace) >= exp.shape[0] * exp.shape[1] // 2 and all(t[4] < 0 for t in trace)
    out = np.zeros_like(exp)
    assert O.me_fullsearch(out, ref, cur, block, window) == 0
    assert np.array_equal(out, exp)
    # the block's answer is the candidate the search stopped at: mv = origin - (left + col, top + row), clamped to +- window / 2, normalised
    for bx, by, col, row, _ in trace:
        ox, oy = bx * block[0], by * block[1]
        left = min(max(ox + block[0] // 2 - 32, 0), w); top = min(max(oy + block[1] // 2 - 32, 0), h)
        mx, my = F(min(max(ox - (left + col), -32), 32)), F(min(max(oy - (top + row), -32), 32))
        want = [O.lib().orc_store_unorm8(float(F(F(mx / F(32)) * F(0.5) + F(0.5)))), 128, O.lib().orc_store_unorm8(float(F(F(my / F(32)) * F(0.5) + F(0.5)))), 255]
        assert out[by, bx].tolist() == want
    # ... and it is not what a search that ignored the exit would return
    assert not np.array_equal(me_python(ref, cur, block, window, (w, h), honour_early_exit=False), exp)
    # a window no taller than the block plus a few rows never gets there
    trace = []
    me_python(*me_frames(1, 64, 48), (16, 16), (24, 24), (64, 48), trace=trace)
    assert trace == []


# ---- GPU ------------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("case", [c[0] for c in snd_cases()])
def test_gpu_snd_matches_oracle(ctx, case):
    from swiftvideo_amd import compute as sv
    _, out0, ins, gains, fades = next(c for c in snd_cases() if c[0] == case)
    u = O.snd_uniforms(gains, fades)
    exp = out0.copy()
    assert O.snd_s16i_s16i(exp, ins, u) == 0
    for misalign in (0, 2):           # 16-byte aligned buffers (128-bit accesses) and buffers starting 2 bytes in (the scalar form)
        n = out0.size
        pad = np.zeros(misalign // 2, dtype=np.int16)
        gout = sv.uploadComputeBuffer(ctx, np.concatenate([pad, out0]).tobytes())
        gins = [sv.uploadComputeBuffer(ctx, np.concatenate([pad, a]).tobytes()) for a in ins]
        K = sv.defaultComputeKernelFromString("snd_s16i_s16i")
        sv.usingContext(ctx, lambda c: sv.runComputeKernel(c, [sv.BufferImage(g, n, offset=misalign) for g in gins],
                                                           sv.BufferImage(gout, n, offset=misalign), K, uniforms=u))
        got = np.frombuffer(sv.downloadComputeBuffer(ctx, gout).tobytes(), dtype=np.int16)[misalign // 2:]
        assert np.array_equal(got, exp), f"{case}, offset {misalign}"


@pytest.mark.gpu
def test_gpu_snd_argument_checks(ctx):
    from swiftvideo_amd import compute as sv
    K = sv.ComputeKernel.snd_s16i_s16i
    buf = sv.uploadComputeBuffer(ctx, np.zeros(64, dtype=np.int16).tobytes())
    short = sv.uploadComputeBuffer(ctx, np.zeros(32, dtype=np.int16).tobytes())
    with pytest.raises(sv.ComputeError):          # fewer buffers than inputCount
        sv.runComputeKernel(ctx, [], sv.BufferImage(buf, 64), K, uniforms=O.snd_uniforms([1.0], [0.5]))
    with pytest.raises(sv.ComputeError):          # an input of another length
        sv.runComputeKernel(ctx, [sv.BufferImage(short, 32)], sv.BufferImage(buf, 64), K, uniforms=O.snd_uniforms([1.0], [0.5]))
    with pytest.raises(sv.ComputeError):          # the ImageUniforms blob is not BufferUniforms
        sv.runComputeKernel(ctx, [sv.BufferImage(buf, 64)], sv.BufferImage(buf, 64), K, uniforms=np.zeros(59, dtype=np.float32))
    # an input that IS the output (or overlaps it): the source accumulates into out[gid] input after input (kernels.cl.swift:548-560), the kernel
    # keeps the accumulator in a register — refused rather than mixed differently
    other = sv.uploadComputeBuffer(ctx, np.ones(64, dtype=np.int16).tobytes())
    with pytest.raises(sv.ComputeError):
        sv.runComputeKernel(ctx, [sv.BufferImage(other, 64), sv.BufferImage(buf, 64)], sv.BufferImage(buf, 64), K, uniforms=O.snd_uniforms([1.0, 1.0], [0.5, 0.5]))
    big = sv.uploadComputeBuffer(ctx, np.zeros(96, dtype=np.int16).tobytes())
    with pytest.raises(sv.ComputeError):          # 64 samples at byte 0 and 64 samples at byte 64 of one buffer
        sv.runComputeKernel(ctx, [sv.BufferImage(big, 64, offset=64)], sv.BufferImage(big, 64), K, uniforms=O.snd_uniforms([1.0], [0.5]))
    sv.runComputeKernel(ctx, [sv.BufferImage(big, 32, offset=64)], sv.BufferImage(big, 32), K, uniforms=O.snd_uniforms([1.0], [0.5]))     # disjoint halves: fine
    sv.endComputePass(ctx, True)


GPU_ME_CASES = ME_CASES + [(11, 320, 180, (16, 16), (32, 32), True), (12, 1920, 1080, (16, 16), (64, 64), False), (13, 200, 120, (64, 64), (64, 64), False),
                           (14, 130, 70, (16, 16), (48, 24), True)]


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(len(GPU_ME_CASES)))
def test_gpu_me_matches_oracle(ctx, case):
    import gpuutil as G
    from swiftvideo_amd import compute as sv
    seed, w, h, block, window, smooth = GPU_ME_CASES[case]
    ref, cur = me_frames(seed, w, h, smooth)
    nbx, nby = -(-w // block[0]), -(-h // block[1])            # every block, also the ones hanging over the right / bottom edge
    exp = np.zeros((nby, nbx, 4), dtype=np.uint8)
    assert O.me_fullsearch(exp, ref, cur, block, window) == 0
    # luma planes: the Y plane of y420p pictures (chroma unused)
    blank = lambda: np.zeros((h // 2 + h % 2, w // 2 + w % 2), dtype=np.uint8)      # noqa: E731
    gref = sv.uploadComputeBuffer(ctx, ref.tobytes())
    gcur = sv.uploadComputeBuffer(ctx, cur.tobytes())
    gout = sv.uploadComputeBuffer(ctx, np.full(nbx * nby * 4, 7, dtype=np.uint8).tobytes())
    u = np.array([block[0], block[1], window[0], window[1], w, h], dtype=np.int32)
    sv.usingContext(ctx, lambda c: sv.runComputeKernel(c, [sv.BufferImage(gref, w, h, 1), sv.BufferImage(gcur, w, h, 1)],
                                                       sv.BufferImage(gout, nbx, nby, 4), sv.ComputeKernel.me_fullsearch, uniforms=u))
    got = sv.downloadComputeBuffer(ctx, gout).reshape(nby, nbx, 4)
    assert np.array_equal(got, exp), f"{np.argwhere((got != exp).any(axis=2))[:5]}"


# ---- committed golden vectors (tests/golden/idle_vectors.npz, tests/golden/gen_idle_golden.py) ---------------------------------------------
from pathlib import Path  # noqa: E402

GOLD = np.load(Path(__file__).resolve().parent / "golden" / "idle_vectors.npz")
GOLD_KEYS = GOLD["index"].tolist()


@pytest.mark.parametrize("key", GOLD_KEYS)
def test_oracle_reproduces_idle_golden_vectors(key):
    exp = GOLD[key + "/expected"]
    if key.startswith("snd"):
        out = GOLD[key + "/out0"].copy()
        ins = [GOLD[f"{key}/in{i}"] for i in range(int(GOLD[key + "/n_inputs"][0]))]
        assert O.snd_s16i_s16i(out, ins, GOLD[key + "/uniforms"]) == 0
    else:
        u = GOLD[key + "/uniforms"].tolist()
        out = np.zeros_like(exp)
        assert O.me_fullsearch(out, GOLD[key + "/ref"], GOLD[key + "/cur"], u[0:2], u[2:4], u[4:6]) == 0
    assert np.array_equal(out, exp)


@pytest.mark.gpu
@pytest.mark.parametrize("key", GOLD_KEYS)
def test_gpu_reproduces_idle_golden_vectors(ctx, key):
    from swiftvideo_amd import compute as sv
    exp = GOLD[key + "/expected"]
    if key.startswith("snd"):
        n = exp.size
        gout = sv.uploadComputeBuffer(ctx, GOLD[key + "/out0"].tobytes())
        gins = [sv.uploadComputeBuffer(ctx, GOLD[f"{key}/in{i}"].tobytes()) for i in range(int(GOLD[key + "/n_inputs"][0]))]
        sv.usingContext(ctx, lambda c: sv.runComputeKernel(c, [sv.BufferImage(g, n) for g in gins], sv.BufferImage(gout, n),
                                                           sv.ComputeKernel.snd_s16i_s16i, uniforms=GOLD[key + "/uniforms"]))
        got = np.frombuffer(sv.downloadComputeBuffer(ctx, gout).tobytes(), dtype=np.int16)
    else:
        ref, cur = GOLD[key + "/ref"], GOLD[key + "/cur"]
        h, w = cur.shape
        gref, gcur = sv.uploadComputeBuffer(ctx, ref.tobytes()), sv.uploadComputeBuffer(ctx, cur.tobytes())
        gout = sv.uploadComputeBuffer(ctx, np.zeros(exp.size, dtype=np.uint8).tobytes())
        sv.usingContext(ctx, lambda c: sv.runComputeKernel(c, [sv.BufferImage(gref, w, h, 1), sv.BufferImage(gcur, w, h, 1)],
                                                           sv.BufferImage(gout, exp.shape[1], exp.shape[0], 4), sv.ComputeKernel.me_fullsearch,
                                                           uniforms=GOLD[key + "/uniforms"]))
        got = sv.downloadComputeBuffer(ctx, gout).reshape(exp.shape)
    assert np.array_equal(got, exp), key
