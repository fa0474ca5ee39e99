"""The device's STORE of strip-kernel geometry tables (csrc/geom_cache.h): tables outlive the batch that built them, keyed by geometry class and
launch configuration, and everything launched through the strip kernels asks the store before its descriptors travel — a LONE tick
(chv_composite: what an unmodified VideoMixer issues, mix.video.swift:95-140) from its third sighting of a scene on, a freshly created batch (a
host that batches builds one per group of frames and runs it once) from the third batch of a scene on.  The tables hold what the set-up code
itself computed, so every launch — first sighting (in place), second (builds), later ones (from the store) — must give the oracle's bytes, with
other pictures behind the same geometry, across strip heights, with animated layers beside static ones, and from two threads at once."""
import threading

import numpy as np
import pytest

import gpuutil as G
import util
from oracle import oracle as O
from swiftvideo_amd import chipvideo as cv
from swiftvideo_amd import compute as sv

pytestmark = pytest.mark.gpu

K = sv.defaultComputeKernelFromString


def _scene(dst, cw, ch, sw, sh, shift=0.0):
    """a video over the whole canvas + two overlays (the mixer's default scene, composer.swift:52-56 in small): [(kernel name, source format, size, uniforms)]"""
    vf = "nv12" if dst == "bgra" else dst
    ovk = "img_bgra_bgra_tx" if dst == "bgra" else f"img_bgra_{dst}"
    return [(f"img_{vf}_{dst}", vf, (sw, sh), util.full_canvas_uniforms((cw, ch), (sw, sh))),
            (ovk, "bgra", (96, 54), util.make_uniforms((cw, ch), rect=(16 + shift, 8, 96, 54), opacity=0.8, in_size=(96, 54))),
            (ovk, "bgra", (96, 54), util.make_uniforms((cw, ch), rect=(cw - 120, ch - 70 + shift, 96, 54), opacity=0.6, in_size=(96, 54)))]


def _tick(ctx, dst, cw, ch, scene, seed):
    """fresh pictures behind `scene`: (target on the device, [(kernel, sample, uniforms, csc)], the oracle's canvas, keepalive)"""
    exp = util.alloc_image(dst, cw, ch, seed=seed)
    assert O.run_kernel(f"img_clear_{dst}", exp) == 0
    layers, keep = [], []
    for l, (name, fmt, (w, h), u) in enumerate(scene):
        src = util.alloc_image(fmt, w, h, seed=seed * 31 + l)
        assert O.run_kernel(name, exp, src, u, threads=4) == 0
        g = G.to_gpu(ctx, fmt, w, h, src)
        keep.append(g)
        layers.append((K(name), g, u, 0))
    gd = G.to_gpu(ctx, dst, cw, ch, util.alloc_image(dst, cw, ch, seed=seed))
    return gd, layers, exp, keep


def _force_strips(switch, dst):
    if dst == "bgra":
        switch("CHV_BGRA_PATH", "wave")
    switch("CHV_YUV_STREAM", "0")


@pytest.mark.parametrize("mode", [None, "eager"])
@pytest.mark.parametrize("dst", ["bgra", "nv12", "y420p"])
def test_lone_ticks_of_a_scene_take_the_stores_tables(ctx, switch, dst, mode):
    """six lone ticks of one scene, new pictures every tick: the first computes in place, the second builds, the rest are pointed at the store's
    tables before their descriptors travel (counter) — all six give the oracle's canvas"""
    _force_strips(switch, dst)
    switch("CHV_GEOM_CACHE", mode)
    cw, ch = 272 + 16 * (mode is None), 160           # (a canvas size of this test's own: the store is the process's)
    scene = _scene(dst, cw, ch, 320, 200)
    p0, b0 = cv.get_counter("geom_store_patched"), cv.get_counter("geom_store_builds")
    for t in range(6):
        gd, layers, exp, keep = _tick(ctx, dst, cw, ch, scene, 100 + t)
        sv.usingContext(ctx, lambda c: sv.compositeTick(c, gd, layers, True))
        G.assert_same(G.from_gpu(ctx, gd, dst, cw, ch), exp, f"{dst} lone tick {t} ({mode})")
    built, patched = cv.get_counter("geom_store_builds") - b0, cv.get_counter("geom_store_patched") - p0
    assert built == 1, built
    assert patched == (5 if mode == "eager" else 4), patched


@pytest.mark.parametrize("dst", ["bgra", "y420p"])
def test_an_animated_layer_beside_static_ones(ctx, switch, dst):
    """one overlay moves every tick (PictureAnimator: a geometry seen once): such ticks find no table for it and compute everything in place — and
    nothing is built on their behalf; when it comes to rest the scene is built and served like any other"""
    _force_strips(switch, dst)
    switch("CHV_GEOM_CACHE", None)
    cw, ch = 304, 176
    b0, p0 = cv.get_counter("geom_store_builds"), cv.get_counter("geom_store_patched")
    for t in range(8):
        scene = _scene(dst, cw, ch, 320, 200, shift=float(min(t, 4)) * 1.25)
        gd, layers, exp, keep = _tick(ctx, dst, cw, ch, scene, 300 + t)
        sv.usingContext(ctx, lambda c: sv.compositeTick(c, gd, layers, True))
        G.assert_same(G.from_gpu(ctx, gd, dst, cw, ch), exp, f"{dst} animated tick {t}")
        if t == 4:
            assert cv.get_counter("geom_store_builds") == b0 and cv.get_counter("geom_store_patched") == p0
    assert cv.get_counter("geom_store_builds") == b0 + 1          # ticks 4 .. 7 are one scene: in place, build, store, store
    assert cv.get_counter("geom_store_patched") == p0 + 2


@pytest.mark.parametrize("dst", ["bgra", "nv12"])
def test_strip_height_is_part_of_the_key(ctx, switch, dst):
    """the same scene with 8-row and 16-row strips in turn: two sets of tables, never one read for the other"""
    _force_strips(switch, dst)
    switch("CHV_GEOM_CACHE", None)
    cw, ch = 336, 192
    scene = _scene(dst, cw, ch, 400, 240)
    b0 = cv.get_counter("geom_store_builds")
    for t in range(10):
        switch("CHV_WAVE_ROWS", "8" if t % 2 == 0 else "16")
        gd, layers, exp, keep = _tick(ctx, dst, cw, ch, scene, 500 + t)
        sv.usingContext(ctx, lambda c: sv.compositeTick(c, gd, layers, True))
        G.assert_same(G.from_gpu(ctx, gd, dst, cw, ch), exp, f"{dst} tick {t}")
    assert cv.get_counter("geom_store_builds") == b0 + 2


@pytest.mark.parametrize("dst", ["bgra", "y420p"])
def test_batches_of_a_scene_start_with_tables_from_the_third_on(ctx, switch, dst):
    """a host that batches: one batch per group of frames, run ONCE, destroyed.  Batch 1 computes in place, batch 2 builds at its only launch and
    gives the tables to the store, batches 3.. are pointed at them at creation — and a lone tick of the scene afterwards is too"""
    _force_strips(switch, dst)
    switch("CHV_GEOM_CACHE", None)
    cw, ch = 368, 208
    scene = _scene(dst, cw, ch, 400, 240)
    b0, p0 = cv.get_counter("geom_store_builds"), cv.get_counter("geom_store_patched")
    for batch in range(5):
        made = [_tick(ctx, dst, cw, ch, scene, 700 + 10 * batch + i) for i in range(4)]
        h, name, ka = G.make_batch(ctx, [(gd, True, layers) for gd, layers, exp, keep in made])
        assert "wave" in name, name
        G.run_batch(ctx, h)
        for i, (gd, layers, exp, keep) in enumerate(made):
            G.assert_same(G.from_gpu(ctx, gd, dst, cw, ch), exp, f"{dst} batch {batch} tick {i} ({name})")
        G.destroy_batch(h)
    assert cv.get_counter("geom_store_builds") == b0 + 1
    assert cv.get_counter("geom_store_patched") == p0 + 3
    gd, layers, exp, keep = _tick(ctx, dst, cw, ch, scene, 799)
    sv.usingContext(ctx, lambda c: sv.compositeTick(c, gd, layers, True))
    G.assert_same(G.from_gpu(ctx, gd, dst, cw, ch), exp, f"{dst} lone tick after the batches")
    assert cv.get_counter("geom_store_patched") == p0 + 4


def test_a_batch_run_many_times_and_the_store(ctx, switch):
    """the bench's shape — one batch run again and again: in place, then built (and given away), then its own pointers; flipping the tables off
    and on again finds them in the store (no second build)"""
    _force_strips(switch, "y420p")
    switch("CHV_GEOM_CACHE", None)
    cw, ch = 400, 224
    scene = _scene("y420p", cw, ch, 400, 240)
    made = [_tick(ctx, "y420p", cw, ch, scene, 900 + i) for i in range(3)]
    h, name, ka = G.make_batch(ctx, [(gd, True, layers) for gd, layers, exp, keep in made])
    b0, h0 = cv.get_counter("geom_store_builds"), cv.get_counter("geom_store_batch_hits")
    for step, val in enumerate([None, None, None, "0", None, None]):
        switch("CHV_GEOM_CACHE", val)
        G.run_batch(ctx, h)
        for i, (gd, layers, exp, keep) in enumerate(made):
            G.assert_same(G.from_gpu(ctx, gd, "y420p", cw, ch), exp, f"step {step} tick {i}")
    G.destroy_batch(h)
    assert cv.get_counter("geom_store_builds") == b0 + 1
    assert cv.get_counter("geom_store_batch_hits") == h0 + 1


def test_two_threads_one_scene(switch):
    """two contexts of one device (VideoMixerGroup: a mixer per queue) tick the same scene at once: whoever builds gives, the other finds — both
    get the oracle's canvases throughout"""
    _force_strips(switch, "nv12")
    switch("CHV_GEOM_CACHE", None)
    cw, ch = 432, 240
    scene = _scene("nv12", cw, ch, 480, 270)
    errors = []

    def run(k):
        try:
            c = sv.makeComputeContext(forType="GPU")
            for t in range(8):
                gd, layers, exp, keep = _tick(c, "nv12", cw, ch, scene, 1100 + 20 * k + t)
                sv.usingContext(c, lambda cc: sv.compositeTick(cc, gd, layers, True))
                G.assert_same(G.from_gpu(c, gd, "nv12", cw, ch), exp, f"thread {k} tick {t}")
        except BaseException as e:           # noqa: BLE001 (reported by the main thread)
            errors.append(e)

    threads = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


def test_held_pass_of_a_scene(ctx, switch):
    """the reference's own call sequence (clear + one kernel per layer inside a pass bracket: held and fused at chv_pass_end) goes through the same
    transient launch and the same store"""
    _force_strips(switch, "y420p")
    switch("CHV_GEOM_CACHE", None)
    cw, ch = 464, 256
    scene = _scene("y420p", cw, ch, 480, 270)
    p0 = cv.get_counter("geom_store_patched")
    for t in range(5):
        gd, layers, exp, keep = _tick(ctx, "y420p", cw, ch, scene, 1300 + t)

        def seq(c):
            c = sv.beginComputePass(c)
            c = sv.runComputeKernel(c, images=[], target=gd, kernel=sv.ComputeKernel.img_clear_y420p, blends=False)
            for k, g, u, csc in layers:
                c = sv.runComputeKernel(c, images=[g], target=gd, kernel=k, uniforms=u, blends=True, colorspace=csc)
            return sv.endComputePass(c, True)
        sv.usingContext(ctx, seq)
        G.assert_same(G.from_gpu(ctx, gd, "y420p", cw, ch), exp, f"held pass {t}")
    assert cv.get_counter("geom_store_patched") == p0 + 3


@pytest.mark.parametrize("dst", ["nv12", "y420p"])
@pytest.mark.parametrize("n_layers", [1, 5, 6, 7, 9])
def test_lone_ticks_carry_their_descriptors_as_kernel_arguments(ctx, switch, dst, n_layers):
    """a lone tick on a 4:2:0 canvas reaches tick_yuv_wave with ticks == nullptr and its descriptors in the kernels' last argument (WaveOne: up to
    six layers; wave_common.hip.h); seven layers and more — and CHV_DESC=device, the A/B — go through the descriptor ring and the device copy.
    Same bytes either way, on first sightings (in place), on the building launch (always through the ring) and on served ones (tables)."""
    _force_strips(switch, dst)
    switch("CHV_GEOM_CACHE", None)
    cw, ch = 208 + 16 * n_layers, 120 + 8 * (dst == "nv12")
    ovk = f"img_bgra_{dst}"
    scene = [(f"img_{dst}_{dst}", dst, (240, 136), util.full_canvas_uniforms((cw, ch), (240, 136)))] + \
            [(ovk, "bgra", (64, 36), util.make_uniforms((cw, ch), rect=(8 + 14 * l, 6 + 9 * l, 64, 36), opacity=0.9 - 0.07 * l, in_size=(64, 36))) for l in range(n_layers - 1)]
    for desc in (None, "device"):
        switch("CHV_DESC", desc)
        for t in range(4):
            gd, layers, exp, keep = _tick(ctx, dst, cw, ch, scene, 1500 + 10 * n_layers + t)
            sv.usingContext(ctx, lambda c: sv.compositeTick(c, gd, layers, True))
            G.assert_same(G.from_gpu(ctx, gd, dst, cw, ch), exp, f"{dst}, {n_layers} layers, CHV_DESC={desc}, tick {t}")


@pytest.mark.parametrize("vf", ["nv12", "y420p"])
@pytest.mark.parametrize("canvas", [(1280, 720), (1472, 960)])
@pytest.mark.parametrize("inset", [False, True])
def test_lone_one_layer_ticks_on_bgra_canvases_by_route(ctx, switch, vf, canvas, inset):
    """a lone tick of ONE video layer on a cleared BGRA canvas: streaming twin below 1.4 Mpixel, strip twin (tick_bgra_wave_one, KINDS 1 / 2) from
    there up and for insets the streaming kernel cannot take (kernels_fast.hip.cpp::select_fast_path) — and the tiled twin and the descriptor ring
    when asked for: the oracle's canvas every way, four ticks each (in place, building, served from the store)"""
    cw, ch = canvas
    sw, sh = 960, 544
    u = util.make_uniforms((cw, ch), rect=(cw // 8, ch // 8, cw // 2, ch // 2), in_size=(sw, sh)) if inset else util.full_canvas_uniforms((cw, ch), (sw, sh))
    name = f"img_{vf}_bgra"
    for route, desc in ((None, None), ("tiled", None), ("wave", None), ("wave", "device")):
        switch("CHV_BGRA_PATH", route)
        switch("CHV_DESC", desc)
        for t in range(4):
            src = util.alloc_image(vf, sw, sh, seed=2000 + t)
            exp = util.alloc_image("bgra", cw, ch, seed=7)
            assert O.run_kernel("img_clear_bgra", exp, threads=8) == 0 and O.run_kernel(name, exp, src, u, threads=8) == 0
            gs = G.to_gpu(ctx, vf, sw, sh, src)
            gd = G.to_gpu(ctx, "bgra", cw, ch, util.alloc_image("bgra", cw, ch, seed=7))
            sv.usingContext(ctx, lambda c: sv.compositeTick(c, gd, [(K(name), gs, u, 0)], True))
            G.assert_same(G.from_gpu(ctx, gd, "bgra", cw, ch), exp, f"{vf} -> {cw}x{ch}, inset {inset}, CHV_BGRA_PATH={route}, CHV_DESC={desc}, tick {t}")
