"""Deferred passes: picture kernels issued between beginComputePass and endComputePass are held by the library and leave as the ONE fused launch
chv_composite would have made of them — what an UNCHANGED VideoMixer issues per tick (mix.video.swift:116-124: clear + N x runComputeKernel,
then endComputePass(wait), compute.cl.swift:234-237,346-359) costs one launch instead of N + 1.  The bytes must be those of the sequence
launched kernel by kernel (CHV_PASS_FUSE=0, or the same calls outside a pass) and of the oracle, for every canvas format."""
import ctypes as C
import gc

import numpy as np
import pytest

import gpuutil as G
import util
from oracle import oracle as O
from swiftvideo_amd import chipvideo as cv
from swiftvideo_amd import compute as sv
from test_gpu_parity import _tick_layers

pytestmark = pytest.mark.gpu

MODES = ["deferred", "immediate_in_pass", "outside_a_pass"]


def _issue(ctx, mode, body):
    """run body(ctx) — a sequence of runComputeKernel calls — the way `mode` says"""
    if mode == "outside_a_pass":
        body(ctx)
        sv.endComputePass(ctx, True)
        return
    if mode == "immediate_in_pass":
        cv.set_switch("CHV_PASS_FUSE", "0")
    try:
        sv.usingContext(ctx, body)
    finally:
        cv.set_switch("CHV_PASS_FUSE", None)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("dst_fmt", ["nv12", "y420p", "bgra"])
def test_the_mixer_sequence_gives_the_oracle_s_bytes_however_it_is_launched(ctx, dst_fmt, mode):
    cw, ch = 64, 36
    layers = _tick_layers(dst_fmt, cw, ch)
    exp = util.alloc_image(dst_fmt, cw, ch, seed=9)
    assert O.run_kernel(f"img_clear_{dst_fmt}", exp) == 0
    srcs = []
    for k, s, iw, ih, u, seed in layers:
        srcs.append(util.alloc_image(s, iw, ih, seed=seed))
        assert O.run_kernel(k, exp, srcs[-1], u) == 0
    canvas = G.to_gpu(ctx, dst_fmt, cw, ch, util.alloc_image(dst_fmt, cw, ch, seed=9))
    k_clear = sv.defaultComputeKernelFromString(f"img_clear_{dst_fmt}")
    up = sv.createComputeContext(sharing=ctx)      # (the upload barrier's context: the mixer's own stays inside its pass)

    def body(c):
        c = sv.runComputeKernel(c, images=[], target=canvas, kernel=k_clear)
        for (k, s, iw, ih, u, _), src in zip(layers, srcs):
            # the source picture is a temporary: its ComputeBuffer's deinit runs when this call returns — before the pass ends
            c = sv.runComputeKernel(c, images=[G.to_gpu(up, s, iw, ih, src)], target=canvas, kernel=sv.defaultComputeKernelFromString(k),
                                    uniforms=u, blends=True)
            gc.collect()
        return c
    _issue(ctx, mode, body)
    sv.destroyComputeContext(up)
    G.assert_same(G.from_gpu(ctx, canvas, dst_fmt, cw, ch), exp, f"{dst_fmt} {mode}")


@pytest.mark.parametrize("mode", MODES)
def test_layers_without_a_clear_continue_on_what_the_canvas_holds(ctx, mode):
    """blends = true reads the current target (compute.cl.swift:296-303): a pass of layer kernels alone composes onto the canvas's bytes"""
    cw, ch = 64, 36
    for dst_fmt in ("bgra", "y420p"):
        layers = _tick_layers(dst_fmt, cw, ch)[1:]
        exp = util.alloc_image(dst_fmt, cw, ch, seed=21)
        srcs = [util.alloc_image(s, iw, ih, seed=seed) for _, s, iw, ih, _, seed in layers]
        for (k, *_rest), src, (_, _, _, _, u, _) in zip(layers, srcs, layers):
            assert O.run_kernel(k, exp, src, u) == 0
        canvas = G.to_gpu(ctx, dst_fmt, cw, ch, util.alloc_image(dst_fmt, cw, ch, seed=21))
        gsrc = [G.to_gpu(ctx, s, iw, ih, src) for (_, s, iw, ih, _, _), src in zip(layers, srcs)]

        def body(c):
            for (k, _, _, _, u, _), g in zip(layers, gsrc):
                c = sv.runComputeKernel(c, images=[g], target=canvas, kernel=sv.defaultComputeKernelFromString(k), uniforms=u, blends=True)
            return c
        _issue(ctx, mode, body)
        G.assert_same(G.from_gpu(ctx, canvas, dst_fmt, cw, ch), exp, f"{dst_fmt} {mode}")


def test_two_canvases_a_clear_in_the_middle_and_a_deep_pass(ctx):
    """one pass: layers on canvas A, a tick on canvas B, back to A with a clear (which discards A's first layers) and 20 more layers — deeper
    than one launch (CHV_MAX_LAYERS = 16).  Issue order is stream order."""
    cw, ch = 64, 36
    la = _tick_layers("bgra", cw, ch)
    lb = _tick_layers("nv12", cw, ch)
    sa = [util.alloc_image(s, iw, ih, seed=seed) for _, s, iw, ih, _, seed in la]
    sb = [util.alloc_image(s, iw, ih, seed=seed + 50) for _, s, iw, ih, _, seed in lb]
    ga = [G.to_gpu(ctx, s, iw, ih, src) for (_, s, iw, ih, _, _), src in zip(la, sa)]
    gb = [G.to_gpu(ctx, s, iw, ih, src) for (_, s, iw, ih, _, _), src in zip(lb, sb)]
    exp_a, exp_b = util.alloc_image("bgra", cw, ch, seed=3), util.alloc_image("nv12", cw, ch, seed=4)
    A, B = G.to_gpu(ctx, "bgra", cw, ch, util.copy_image(exp_a)), G.to_gpu(ctx, "nv12", cw, ch, util.copy_image(exp_b))
    K = sv.defaultComputeKernelFromString
    deep = [(i * 7) % 4 for i in range(20)]

    def run(on, canvas, layers, gpu, srcs, i, c):
        k, _, _, _, u, _ = layers[i]
        if on == "oracle":
            assert O.run_kernel(k, canvas, srcs[i], u) == 0
            return c
        return sv.runComputeKernel(c, images=[gpu[i]], target=canvas, kernel=K(k), uniforms=u, blends=True)

    def sequence(on, a, b, c=None):
        c = run(on, a, la, ga, sa, 1, c)
        c = run(on, a, la, ga, sa, 2, c)
        if on == "oracle":
            assert O.run_kernel("img_clear_nv12", b) == 0
        else:
            c = sv.runComputeKernel(c, images=[], target=b, kernel=K("img_clear_nv12"))
        for i in range(4):
            c = run(on, b, lb, gb, sb, i, c)
        if on == "oracle":
            assert O.run_kernel("img_clear_bgra", a) == 0
        else:
            c = sv.runComputeKernel(c, images=[], target=a, kernel=K("img_clear_bgra"))
        for i in deep:
            c = run(on, a, la, ga, sa, i, c)
        return c
    sequence("oracle", exp_a, exp_b)
    sv.usingContext(ctx, lambda c: sequence("gpu", A, B, c))
    G.assert_same(G.from_gpu(ctx, A, "bgra", cw, ch), exp_a, "canvas A")
    G.assert_same(G.from_gpu(ctx, B, "nv12", cw, ch), exp_b, "canvas B")


def test_a_layer_that_reads_the_canvas_held_before_it_sees_what_was_issued_before_it(ctx):
    """pass: clear A, X -> A, then A -> B and B -> A (a picture-in-picture of the mix so far, and back): every kernel samples what the kernels issued
    before it wrote — the held canvas goes out before anything reads it, whether as another target's source or as its own"""
    cw, ch = 64, 36
    x = util.alloc_image("nv12", 96, 54, seed=31)
    ux = util.full_canvas_uniforms((cw, ch), (96, 54))
    uab = util.make_uniforms((cw, ch), rect=(8, 4, 40, 24), opacity=0.7, in_size=(cw, ch))
    uba = util.make_uniforms((cw, ch), rect=(20, 10, 30, 20), opacity=0.9, in_size=(cw, ch))
    ea, eb = util.alloc_image("bgra", cw, ch, seed=32), util.alloc_image("bgra", cw, ch, seed=33)
    A, B = G.to_gpu(ctx, "bgra", cw, ch, util.copy_image(ea)), G.to_gpu(ctx, "bgra", cw, ch, util.copy_image(eb))
    gx = G.to_gpu(ctx, "nv12", 96, 54, x)
    assert O.run_kernel("img_clear_bgra", ea) == 0 and O.run_kernel("img_nv12_bgra", ea, x, ux) == 0
    assert O.run_kernel("img_bgra_bgra_tx", eb, ea, uab) == 0            # B <- A as composed so far
    assert O.run_kernel("img_nv12_bgra", eb, x, util.full_canvas_uniforms((cw, ch), (96, 54), opacity=0.25)) == 0
    assert O.run_kernel("img_bgra_bgra_tx", ea, eb, uba) == 0            # A <- B, after B's second layer
    K = sv.ComputeKernel

    def body(c):
        c = sv.runComputeKernel(c, images=[], target=A, kernel=K.img_clear_bgra)
        c = sv.runComputeKernel(c, images=[gx], target=A, kernel=K.img_nv12_bgra, uniforms=ux, blends=True)
        c = sv.runComputeKernel(c, images=[A], target=B, kernel=K.img_bgra_bgra_tx, uniforms=uab, blends=True)
        c = sv.runComputeKernel(c, images=[gx], target=B, kernel=K.img_nv12_bgra, uniforms=util.full_canvas_uniforms((cw, ch), (96, 54), opacity=0.25), blends=True)
        return sv.runComputeKernel(c, images=[B], target=A, kernel=K.img_bgra_bgra_tx, uniforms=uba, blends=True)
    sv.usingContext(ctx, body)
    G.assert_same(G.from_gpu(ctx, A, "bgra", cw, ch), ea, "canvas A")
    G.assert_same(G.from_gpu(ctx, B, "bgra", cw, ch), eb, "canvas B")


def test_an_argument_error_comes_back_from_its_own_call_and_the_pass_goes_on(ctx):
    cw, ch = 64, 36
    src = util.alloc_image("nv12", 96, 54, seed=5)
    u = util.full_canvas_uniforms((cw, ch), (96, 54))
    exp = util.alloc_image("bgra", cw, ch, seed=6)
    assert O.run_kernel("img_clear_bgra", exp) == 0 and O.run_kernel("img_nv12_bgra", exp, src, u) == 0
    canvas, g = G.to_gpu(ctx, "bgra", cw, ch, util.alloc_image("bgra", cw, ch, seed=6)), G.to_gpu(ctx, "nv12", 96, 54, src)
    bgra = G.to_gpu(ctx, "bgra", 16, 8, util.alloc_image("bgra", 16, 8, seed=1))
    sv.beginComputePass(ctx)
    sv.runComputeKernel(ctx, images=[], target=canvas, kernel=sv.ComputeKernel.img_clear_bgra)
    with pytest.raises(sv.ComputeError):          # BGRA planes handed to an NV12 kernel: refused here, not at endComputePass
        sv.runComputeKernel(ctx, images=[bgra], target=canvas, kernel=sv.ComputeKernel.img_nv12_bgra, uniforms=u, blends=True)
    with pytest.raises(sv.ComputeError):          # a layer kernel that does not blend
        sv.runComputeKernel(ctx, images=[g], target=canvas, kernel=sv.ComputeKernel.img_nv12_bgra, uniforms=u, blends=False)
    sv.runComputeKernel(ctx, images=[g], target=canvas, kernel=sv.ComputeKernel.img_nv12_bgra, uniforms=u, blends=True)
    sv.endComputePass(ctx, True)
    G.assert_same(G.from_gpu(ctx, canvas, "bgra", cw, ch), exp, "after two refused kernels")


def test_other_work_on_the_context_keeps_its_place_behind_the_held_kernels(ctx):
    """a download, an upload into a source, a batch and an event in the middle of a pass: each first sends out what the pass holds"""
    cw, ch = 64, 36
    s1, s2 = util.alloc_image("nv12", 96, 54, seed=11), util.alloc_image("nv12", 96, 54, seed=12)
    u = util.full_canvas_uniforms((cw, ch), (96, 54))
    u2 = util.full_canvas_uniforms((cw, ch), (96, 54), opacity=0.5)
    canvas = G.to_gpu(ctx, "bgra", cw, ch, util.alloc_image("bgra", cw, ch, seed=13))
    g = G.to_gpu(ctx, "nv12", 96, 54, s1)
    step1 = util.alloc_image("bgra", cw, ch, seed=13)
    assert O.run_kernel("img_clear_bgra", step1) == 0 and O.run_kernel("img_nv12_bgra", step1, s1, u) == 0
    step2 = util.copy_image(step1)
    assert O.run_kernel("img_nv12_bgra", step2, s2, u2) == 0
    lib = cv.load()
    sv.beginComputePass(ctx)
    sv.runComputeKernel(ctx, images=[], target=canvas, kernel=sv.ComputeKernel.img_clear_bgra)
    sv.runComputeKernel(ctx, images=[g], target=canvas, kernel=sv.ComputeKernel.img_nv12_bgra, uniforms=u, blends=True)
    # the download sees clear + layer 1 although the pass has not ended
    G.assert_same(G.from_gpu(ctx, canvas, "bgra", cw, ch), step1, "download inside the pass")
    # layer 2 reads the SAME device picture after new bytes were uploaded into it: the upload must not overtake ... nothing is held here,
    # but the kernel accepted next must see the new bytes, and the one accepted before must not
    sv.runComputeKernel(ctx, images=[g], target=canvas, kernel=sv.ComputeKernel.img_nv12_bgra, uniforms=u2, blends=True)     # (held: reads s1)
    img = g.imageBuffer()
    cpu2 = sv.pictureFromArrays(sv.PixelFormat.nv12, (96, 54), s2).imageBuffer()
    for off, pitch, src, src_pitch, wb, rows in sv._upload_regions(cpu2, img.gpuPitches, img.gpuOffsets):
        cv.check(lib.chv_upload(ctx.handle, img.computeTextures[0]._h, off, pitch, src, src_pitch, wb, rows, 0))
    ev = C.c_void_p()
    cv.check(lib.chv_event_create(ctx.handle, C.byref(ev)))
    cv.check(lib.chv_event_record(ctx.handle, ev))
    sv.endComputePass(ctx, True)
    cv.check(lib.chv_event_destroy(ev))
    # layer 2 was accepted BEFORE the upload: it composed s1 (at half opacity), not s2
    want = util.copy_image(step1)
    assert O.run_kernel("img_nv12_bgra", want, s1, u2) == 0
    G.assert_same(G.from_gpu(ctx, canvas, "bgra", cw, ch), want, "layer accepted before the upload")
    # ... and a kernel accepted after it composes the new bytes
    sv.usingContext(ctx, lambda c: sv.runComputeKernel(
        sv.runComputeKernel(c, images=[], target=canvas, kernel=sv.ComputeKernel.img_clear_bgra),
        images=[g], target=canvas, kernel=sv.ComputeKernel.img_nv12_bgra, uniforms=u, blends=True))
    want = util.alloc_image("bgra", cw, ch, seed=13)
    assert O.run_kernel("img_clear_bgra", want) == 0 and O.run_kernel("img_nv12_bgra", want, s2, u) == 0
    G.assert_same(G.from_gpu(ctx, canvas, "bgra", cw, ch), want, "kernel accepted after the upload")


@pytest.mark.parametrize("seed", range(12))
def test_random_passes_deferred_equals_kernel_by_kernel(ctx, seed):
    """random sequences of clears and layers on two canvases of random formats: held-and-fused == launched one by one, byte for byte"""
    rng = np.random.default_rng(900 + seed)
    cw, ch = 64, 36
    fmts = [str(rng.choice(["bgra", "nv12", "y420p"])) for _ in range(2)]
    pools = [_tick_layers(f, cw, ch) for f in fmts]
    srcs = [[util.alloc_image(s, iw, ih, seed=sd + 7 * seed) for _, s, iw, ih, _, sd in p] for p in pools]
    gpu = [[G.to_gpu(ctx, s, iw, ih, src) for (_, s, iw, ih, _, _), src in zip(p, ss)] for p, ss in zip(pools, srcs)]
    ops = []
    for _ in range(int(rng.integers(3, 30))):
        t = int(rng.integers(0, 2))
        ops.append((t, -1) if rng.random() < 0.15 else (t, int(rng.integers(0, 4))))
    K = sv.defaultComputeKernelFromString
    results = []
    for mode in ("deferred", "immediate_in_pass"):
        canvases = [G.to_gpu(ctx, f, cw, ch, util.alloc_image(f, cw, ch, seed=70 + i)) for i, f in enumerate(fmts)]

        def body(c):
            for t, i in ops:
                if i < 0:
                    c = sv.runComputeKernel(c, images=[], target=canvases[t], kernel=K(f"img_clear_{fmts[t]}"))
                else:
                    k, _, _, _, u, _ = pools[t][i]
                    c = sv.runComputeKernel(c, images=[gpu[t][i]], target=canvases[t], kernel=K(k), uniforms=u, blends=True)
            return c
        _issue(ctx, mode, body)
        results.append([G.from_gpu(ctx, cn, f, cw, ch) for cn, f in zip(canvases, fmts)])
    for a, b, f in zip(results[0], results[1], fmts):
        G.assert_same(a, b, f"{f}, ops {ops}")
    # and the oracle
    exp = [util.alloc_image(f, cw, ch, seed=70 + i) for i, f in enumerate(fmts)]
    for t, i in ops:
        if i < 0:
            assert O.run_kernel(f"img_clear_{fmts[t]}", exp[t]) == 0
        else:
            k, _, _, _, u, _ = pools[t][i]
            assert O.run_kernel(k, exp[t], srcs[t][i], u) == 0
    for a, e, f in zip(results[0], exp, fmts):
        G.assert_same(a, e, f"{f} vs oracle, ops {ops}")
