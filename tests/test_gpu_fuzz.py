"""Seeded random geometry/format/size sweeps: HIP path == oracle, bit for bit; and concurrent use of
several contexts from several threads (the way Bus runner threads enter the backend, compute.swift:177,234)."""
import threading

import numpy as np
import pytest

import gpuutil as G
import scenarios as S
import util
from oracle import oracle as O
from swiftvideo_amd import compute as sv

pytestmark = pytest.mark.gpu
KERNELS = S.LAYER_KERNELS_REF + S.LAYER_KERNELS_OWN


def _random_case(rng):
    cw, ch = int(rng.integers(2, 150)), int(rng.integers(2, 90))
    iw, ih = int(rng.integers(2, 200)), int(rng.integers(2, 120))
    kw = {}
    if rng.random() < 0.8:
        kw["rect"] = (float(rng.uniform(-0.4, 0.8) * cw), float(rng.uniform(-0.4, 0.8) * ch),
                      float(rng.uniform(0.1, 1.6) * cw), float(rng.uniform(0.1, 1.6) * ch))
    if rng.random() < 0.4:
        kw["rotation"] = float(rng.uniform(-3.2, 3.2))
    if rng.random() < 0.5:
        kw["border"] = tuple(float(v) for v in rng.uniform(0, 12, 4))
    if rng.random() < 0.5:
        kw["fill"] = tuple(float(v) for v in rng.uniform(0, 1, 4))
    if rng.random() < 0.5:
        kw["tex"] = (float(rng.uniform(-0.2, 0.5)), float(rng.uniform(-0.2, 0.5)), float(rng.uniform(0.2, 1.3)) * (1 if rng.random() < 0.8 else -1),
                     float(rng.uniform(0.2, 1.3)))
    kw["opacity"] = float(rng.choice([1.0, 0.0, rng.uniform(0, 1)]))
    return cw, ch, iw, ih, kw


@pytest.mark.parametrize("seed", range(40))
def test_random_layer_matches_oracle(ctx, seed):
    rng = np.random.default_rng(1000 + seed)
    kernel = KERNELS[seed % len(KERNELS)]
    cw, ch, iw, ih, kw = _random_case(rng)
    s, d = G.kernel_formats(kernel)
    if d != "bgra":
        cw, ch = max(cw, 2), max(ch, 2)
    if s in ("nv12", "y420p"):
        iw, ih = max(iw, 2), max(ih, 2)
    u = util.make_uniforms((cw, ch), in_size=(iw, ih), **kw)
    clear = bool(rng.integers(0, 2))
    got, exp = G.run_both(ctx, kernel, cw, ch, iw, ih, u, seed=5000 + seed, csc=int(rng.integers(0, 4)), clear_first=clear)
    G.assert_same(got, exp, f"{kernel} seed {seed} {cw}x{ch} <- {iw}x{ih} {kw}")


@pytest.mark.parametrize("seed", range(8))
def test_random_ticks_batched(ctx, seed):
    """Random multi-layer ticks of one target format in one batch vs the oracle's per-layer sequence."""
    rng = np.random.default_rng(2000 + seed)
    d = ["nv12", "y420p", "bgra"][seed % 3]
    srcs_for = {"nv12": ["nv12", "y420p", "bgra", "rgba"], "y420p": ["y420p", "bgra", "rgba"], "bgra": ["nv12", "y420p", "bgra", "rgba"]}[d]
    ticks, exps, gds = [], [], []
    for t in range(5):
        cw, ch = int(rng.integers(2, 100)) * 2, int(rng.integers(2, 50)) * 2
        exp = util.alloc_image(d, cw, ch, seed=1)
        assert O.run_kernel(f"img_clear_{d}", exp) == 0
        layers = []
        for l in range(int(rng.integers(0, 5))):
            s = srcs_for[int(rng.integers(0, len(srcs_for)))]
            name = f"img_{s}_{d}" + ("_tx" if d == "bgra" and s in ("bgra", "rgba") else "")
            if name == "img_nv12_y420p":
                continue
            _, _, iw, ih, kw = _random_case(rng)
            iw, ih = max(iw, 2), max(ih, 2)
            u = util.make_uniforms((cw, ch), in_size=(iw, ih), **kw)
            src = util.alloc_image(s, iw, ih, seed=int(rng.integers(1, 1 << 20)))
            assert O.run_kernel(name, exp, src, u, threads=2) == 0
            layers.append((sv.defaultComputeKernelFromString(name), G.to_gpu(ctx, s, iw, ih, src), u, 0))
        gd = G.to_gpu(ctx, d, cw, ch, util.alloc_image(d, cw, ch, seed=1))
        ticks.append((gd, True, layers))
        exps.append(exp)
        gds.append((gd, cw, ch))
    h, name, keep = G.make_batch(ctx, ticks)
    G.run_batch(ctx, h)
    G.destroy_batch(h)
    for i, ((gd, cw, ch), exp) in enumerate(zip(gds, exps)):
        G.assert_same(G.from_gpu(ctx, gd, d, cw, ch), exp, f"seed {seed} tick {i} via {name}")


def test_concurrent_contexts_from_threads(ctx):
    """Four threads, each with its own sharing context, upload + composite + download concurrently."""
    errors = []

    def worker(i):
        try:
            c = sv.createComputeContext(sharing=ctx)
            rng = np.random.default_rng(300 + i)
            for it in range(6):
                cw, ch, iw, ih = 256 + 16 * i, 96, 192, 108
                src = util.alloc_image("nv12", iw, ih, seed=int(rng.integers(1, 1 << 20)))
                u = util.make_uniforms((cw, ch), in_size=(iw, ih), opacity=float(rng.uniform(0.2, 1.0)))
                exp = util.alloc_image("bgra", cw, ch)
                O.run_kernel("img_clear_bgra", exp)
                O.run_kernel("img_nv12_bgra", exp, src, u)
                gs = sv.uploadComputePicture(c, sv.pictureFromArrays(sv.PixelFormat.nv12, (iw, ih), src), asynchronous=bool(it % 2))
                gd = sv.uploadComputePicture(c, sv.createPictureSample((cw, ch), sv.PixelFormat.BGRA))
                sv.usingContext(c, lambda cc: sv.compositeTick(cc, gd, [(sv.ComputeKernel.img_nv12_bgra, gs, u, 0)], True))
                got = G.from_gpu(c, gd, "bgra", cw, ch)
                G.assert_same(got, exp, f"thread {i} iteration {it}")
            sv.destroyComputeContext(c)
        except Exception as e:  # noqa: BLE001
            errors.append(f"thread {i}: {e}")

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


# ---- forced-route equivalence over further seeds (the fuzzers that used to live under tools/ and were run by hand) --------------------
# Every kernel family that can take a tick is FORCED to take it and held to the oracle's bytes: a regression in one route cannot hide behind
# the default route choosing another.  The seeded generators are the ones of the per-family test files; the seed ranges continue theirs.
import test_gpu_mixpath as MIX  # noqa: E402
import test_gpu_parity as PAR  # noqa: E402
import test_gpu_yuvstream as YST  # noqa: E402
import test_gpu_yuvwave as YWV  # noqa: E402


@pytest.mark.parametrize("seed", range(36, 90))
def test_fuzz_bgra_stream_route(ctx, switch, seed):
    """tick_bgra_stream forced (CHV_BGRA_PATH=stream), batched; then the same ticks one at a time (descriptors as kernel arguments)"""
    MIX.test_random_stream_ticks(ctx, switch, seed)
    switch("CHV_BGRA_PATH", None)
    MIX.test_random_lone_stream_ticks(ctx, seed)


@pytest.mark.parametrize("rows", ["8", "16"])
@pytest.mark.parametrize("seed", range(100, 126))
def test_fuzz_strip_routes(ctx, switch, rows, seed):
    """tick_bgra_wave and tick_yuv_wave forced, both strip heights: strong reductions (the pair form), flips, borders, fill, layers across edges"""
    switch("CHV_BGRA_PATH", "wave")
    switch("CHV_YUV_STREAM", "0")
    switch("CHV_WAVE_ROWS", rows)
    MIX.test_random_mixed_ticks(ctx, MIX.WAVE, seed)
    YWV.test_random_yuv_ticks(ctx, rows, seed)


@pytest.mark.parametrize("seed", range(24, 40))
def test_fuzz_rgb_only_strips_without_dma_staging(ctx, switch, seed):
    """the RGB-only instantiation of tick_bgra_wave with its LDS-DMA staging switched OFF (CHV_WAVE_DMA=0 through chv_debug_set_switch: the switch
    used to be read from the environment once, so no in-process test could reach the register staging of that instantiation) — and ON, on the
    same seeds"""
    switch("CHV_WAVE_DMA", "0")
    MIX.test_random_rgb_only_ticks(ctx, MIX.WAVE, seed)
    switch("CHV_WAVE_DMA", None)
    MIX.test_random_rgb_only_ticks(ctx, MIX.WAVE, seed)


@pytest.mark.parametrize("seed", range(200, 252))
def test_fuzz_yuv_stream_route(ctx, switch, seed):
    """tick_yuv_stream forced for every eligible launch: float and integer-matrix RGB layers, batched and lone"""
    switch("CHV_YUV_STREAM", "force")
    YST.test_random_yuv_stream_ticks(ctx, seed)
    YST.test_random_yuv_stream_ticks(ctx, f"int{seed}")
    YST.test_random_lone_yuv_stream_ticks(ctx, seed)


@pytest.mark.parametrize("seed", range(48, 100))
def test_fuzz_lanczos(ctx, seed):
    PAR.test_lanczos_random_geometries(ctx, seed)


@pytest.mark.parametrize("seed", range(40, 60))
def test_fuzz_general_route(ctx, switch, seed):
    """the general kernels forced (CHV_FORCE_GENERAL=1) on the strip kernels' random ticks: the route everything else falls back to"""
    switch("CHV_FORCE_GENERAL", "1")
    MIX.test_random_mixed_ticks(ctx, None, seed)
    YWV.test_random_yuv_ticks(ctx, "8", seed)
