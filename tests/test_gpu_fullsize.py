"""BASELINE.json's configurations at full size: direct comparison with the (multi-threaded) oracle where
it finishes in seconds, plus size-independent properties (idempotent replay, fused == sequential,
opacity-0 identity, constant images stay constant, replication of a frame gives replicated outputs)."""
import os

import numpy as np
import pytest

import gpuutil as G
import util
from oracle import oracle as O
from swiftvideo_amd import compute as sv

pytestmark = pytest.mark.gpu
CORES = os.cpu_count() or 4


def test_cfg1_720p_nv12_to_bgra(ctx):
    """configs[0]: single 1280x720 NV12 -> BGRA convert (no scaling)."""
    w, h = 1280, 720
    src = util.alloc_image("nv12", w, h, seed=0x5EED0000 + 16)
    u = util.full_canvas_uniforms((w, h), (w, h))
    exp = util.alloc_image("bgra", w, h)
    assert O.run_kernel("img_clear_bgra", exp, threads=CORES) == 0
    assert O.run_kernel("img_nv12_bgra", exp, src, u, threads=CORES) == 0
    gs, gd = G.to_gpu(ctx, "nv12", w, h, src), G.to_gpu(ctx, "bgra", w, h, util.alloc_image("bgra", w, h, seed=1))
    sv.usingContext(ctx, lambda c: sv.compositeTick(c, gd, [(sv.ComputeKernel.img_nv12_bgra, gs, u, 0)], True))
    G.assert_same(G.from_gpu(ctx, gd, "bgra", w, h), exp, "cfg1")


@pytest.mark.parametrize("csc", [0, 1])
def test_cfg2_1080p_nv12_to_720p_bgra(ctx, csc):
    """configs[1]: 1920x1080 NV12 -> BGRA + bilinear downscale to 1280x720 (the bench workload)."""
    sw, sh, dw, dh = 1920, 1080, 1280, 720
    src = util.alloc_image("nv12", sw, sh, seed=0x5EED0000 + 32)
    u = util.full_canvas_uniforms((dw, dh), (sw, sh))
    exp = util.alloc_image("bgra", dw, dh)
    assert O.run_kernel("img_clear_bgra", exp, threads=CORES) == 0
    assert O.run_kernel("img_nv12_bgra", exp, src, u, csc=csc, threads=CORES) == 0
    gs = G.to_gpu(ctx, "nv12", sw, sh, src)
    gd = G.to_gpu(ctx, "bgra", dw, dh, util.alloc_image("bgra", dw, dh, seed=2))
    h, name, keep = G.make_batch(ctx, [(gd, True, [(sv.ComputeKernel.img_nv12_bgra, gs, u, csc)])])
    assert name == "tick_nv12_bgra_tiled"
    G.run_batch(ctx, h)
    first = G.from_gpu(ctx, gd, "bgra", dw, dh)
    G.assert_same(first, exp, "cfg2")
    G.run_batch(ctx, h)                                   # idempotent replay
    G.assert_same(G.from_gpu(ctx, gd, "bgra", dw, dh), exp, "cfg2 replay")
    G.destroy_batch(h)


def test_cfg3_four_1080p_bgra_layers(ctx):
    """configs[2]: VideoMixer with four 1080p BGRA layers (opacity 1, .75, .5, .25) onto a 1080p canvas, with the
    transform- and opacity-aware kernel family (bgraKernelFamily="tx": the optional mix.video.swift hunk of INTEGRATION.md)."""
    w, h = 1920, 1080
    layers = [util.alloc_image("bgra", w, h, seed=0x5EED0000 + 48 + i) for i in range(4)]
    us = [util.full_canvas_uniforms((w, h), (w, h), opacity=o) for o in (1.0, 0.75, 0.5, 0.25)]
    exp = util.alloc_image("bgra", w, h)
    assert O.run_kernel("img_clear_bgra", exp, threads=CORES) == 0
    for l, u in zip(layers, us):
        assert O.run_kernel("img_bgra_bgra_tx", exp, l, u, threads=CORES) == 0
    mixer = sv.VideoMixer("ws", 1 / 30, (w, h), outputFormat=sv.PixelFormat.BGRA, computeContext=ctx, fused=True, bgraKernelFamily="tx")
    up = sv.GPUBarrierUpload(ctx)
    M = util.ortho(w, h) @ util._mat_scale(w, h)
    for z, (l, o) in enumerate(zip(layers, (1.0, 0.75, 0.5, 0.25))):
        tag, s = up(sv.pictureFromArrays(sv.PixelFormat.BGRA, (w, h), l, matrix=M, opacity=o, zIndex=z, assetId=f"in{z}"))
        assert tag == "just"
        assert mixer.push(s)[0] == "nothing"
    out = mixer.mix(at=0.0)
    assert out is not None, mixer.result
    G.assert_same(G.from_gpu(ctx, out, "bgra", w, h), exp, "cfg3 fused mixer tick")
    # the reference's own call sequence (clear + 4 x applyComputeImage) gives the same bytes
    mixer2 = sv.VideoMixer("ws", 1 / 30, (w, h), outputFormat=sv.PixelFormat.BGRA, computeContext=ctx, fused=False, bgraKernelFamily="tx")
    for z, (l, o) in enumerate(zip(layers, (1.0, 0.75, 0.5, 0.25))):
        mixer2.push(up(sv.pictureFromArrays(sv.PixelFormat.BGRA, (w, h), l, matrix=M, opacity=o, zIndex=z, assetId=f"in{z}"))[1])
    out2 = mixer2.mix(at=0.0)
    G.assert_same(G.from_gpu(ctx, out2, "bgra", w, h), exp, "cfg3 sequential mixer tick")


def test_default_kernel_family_tick_with_a_bgra_overlay(ctx):
    """What an UNCHANGED VideoMixer issues on a 1080p BGRA canvas (findKernel, mix.video.swift:167-182; no `bgraKernelFamily` opt-in):
    two 1080p NV12 videos through img_nv12_bgra and a 1080p BGRA overlay through img_bgra_bgra (kernels.metal:52-62).  The video layers
    keep the strip kernel (the overlay is applied per pixel inside it); fused tick == the reference's call sequence == the oracle."""
    w, h = 1920, 1080
    vids = [util.alloc_image("nv12", w, h, seed=0x5EED0000 + 112 + i) for i in range(2)]
    logo = util.alloc_image("bgra", w, h, seed=0x5EED0000 + 120)
    M = util.ortho(w, h) @ util._mat_scale(w, h)
    exp = util.alloc_image("bgra", w, h)
    assert O.run_kernel("img_clear_bgra", exp, threads=CORES) == 0
    for v, o in zip(vids, (1.0, 0.5)):
        assert O.run_kernel("img_nv12_bgra", exp, v, util.full_canvas_uniforms((w, h), (w, h), opacity=o), threads=CORES) == 0
    assert O.run_kernel("img_bgra_bgra", exp, logo, util.full_canvas_uniforms((w, h), (w, h)), threads=CORES) == 0
    up = sv.GPUBarrierUpload(ctx)
    for fused in (True, False):
        mixer = sv.VideoMixer("ws", 1 / 30, (w, h), outputFormat=sv.PixelFormat.BGRA, computeContext=ctx, fused=fused)
        pics = [sv.pictureFromArrays(sv.PixelFormat.nv12, (w, h), v, matrix=M, opacity=o, zIndex=z, assetId=f"v{z}")
                for z, (v, o) in enumerate(zip(vids, (1.0, 0.5)))]
        pics.append(sv.pictureFromArrays(sv.PixelFormat.BGRA, (w, h), logo, matrix=M, zIndex=2, assetId="logo"))
        for p in pics:
            assert mixer.push(up(p)[1])[0] == "nothing"
        out = mixer.mix(at=0.0)
        assert out is not None, mixer.result
        G.assert_same(G.from_gpu(ctx, out, "bgra", w, h), exp, f"default family, fused={fused}")
    # and the launches a batch of this tick is: the streaming kernel for the two videos, the strip kernel — not the general one — for the overlay
    # (chv_batch_create splits "videos of one geometry, then something else")
    gv = [G.to_gpu(ctx, "nv12", w, h, v) for v in vids]
    gl = G.to_gpu(ctx, "bgra", w, h, logo)
    gd = G.to_gpu(ctx, "bgra", w, h, util.alloc_image("bgra", w, h, seed=3))
    u = util.full_canvas_uniforms((w, h), (w, h))
    layers = [(sv.ComputeKernel.img_nv12_bgra, gv[0], u, 0),
              (sv.ComputeKernel.img_nv12_bgra, gv[1], util.full_canvas_uniforms((w, h), (w, h), opacity=0.5), 0),
              (sv.ComputeKernel.img_bgra_bgra, gl, u, 0)]
    hb, name, keep = G.make_batch(ctx, [(gd, True, layers)])
    assert name == "tick_bgra_stream + tick_bgra_wave", name
    G.run_batch(ctx, hb)
    G.destroy_batch(hb)
    G.assert_same(G.from_gpu(ctx, gd, "bgra", w, h), exp, "default family as one batch")


def test_cfg4_streams_are_independent(ctx):
    """configs[3] at its STATED stream count on one device: 64 concurrent 1080p NV12 -> 720p BGRA streams in one launch (eight times what a GPU
    of the 8-GPU line gets), four distinct sources, every stream with a source picture and a canvas of its own; every stream's output equals
    that stream's single-tick oracle result (no cross-talk between the ticks of a batch), and the batch run a second time — the next tick of
    every bus — gives the same bytes."""
    sw, sh, dw, dh = 1920, 1080, 1280, 720
    n_streams, distinct = 64, 4
    u = util.full_canvas_uniforms((dw, dh), (sw, sh))
    srcs = [util.alloc_image("nv12", sw, sh, seed=0x5EED0000 + 64 + s) for s in range(distinct)]
    exps = []
    for s in srcs:
        e = util.alloc_image("bgra", dw, dh)
        assert O.run_kernel("img_clear_bgra", e, threads=CORES) == 0
        assert O.run_kernel("img_nv12_bgra", e, s, u, threads=CORES) == 0
        exps.append(e)
    ticks, gds = [], []
    for i in range(n_streams):
        gs = G.to_gpu(ctx, "nv12", sw, sh, srcs[i % distinct])
        gd = G.to_gpu(ctx, "bgra", dw, dh, util.alloc_image("bgra", dw, dh, seed=70 + i))
        ticks.append((gd, True, [(sv.ComputeKernel.img_nv12_bgra, gs, u, 0)]))
        gds.append(gd)
    h, name, keep = G.make_batch(ctx, ticks)
    for rerun in range(2):
        G.run_batch(ctx, h)
        for i, gd in enumerate(gds):
            G.assert_same(G.from_gpu(ctx, gd, "bgra", dw, dh), exps[i % distinct], f"stream {i} of {n_streams} ({name}), run {rerun}")
    G.destroy_batch(h)


def test_cfg5_4k_eight_layers_then_lanczos(ctx):
    """configs[4]: 8 x 3840x2160 BGRA layers composited onto a 2160p canvas, Lanczos-3 down to 1080p."""
    w, h, ow, oh = 3840, 2160, 1920, 1080
    base = [util.alloc_image("bgra", w, h, seed=0x5EED0000 + 80 + i) for i in range(2)]
    ops = (1.0, 0.9, 0.8, 0.7, 0.6, 0.5, 0.4, 0.3)
    us = [util.full_canvas_uniforms((w, h), (w, h), opacity=o) for o in ops]
    exp = util.alloc_image("bgra", w, h)
    assert O.run_kernel("img_clear_bgra", exp, threads=CORES) == 0
    for i, u in enumerate(us):
        assert O.run_kernel("img_bgra_bgra_tx", exp, base[i % 2], u, threads=CORES) == 0
    exp_small = util.alloc_image("bgra", ow, oh)
    assert O.lanczos_bgra(exp_small[0], exp[0], threads=CORES) == 0
    gl = [G.to_gpu(ctx, "bgra", w, h, b) for b in base]
    canvas = G.to_gpu(ctx, "bgra", w, h, util.alloc_image("bgra", w, h))
    small = G.to_gpu(ctx, "bgra", ow, oh, util.alloc_image("bgra", ow, oh))
    layers = [(sv.ComputeKernel.img_bgra_bgra_tx, gl[i % 2], u, 0) for i, u in enumerate(us)]

    def body(c):
        c = sv.compositeTick(c, canvas, layers, True)
        return sv.scaleLanczos(c, small, canvas)
    sv.usingContext(ctx, body)
    G.assert_same(G.from_gpu(ctx, canvas, "bgra", w, h), exp, "cfg5 composite")
    G.assert_same(G.from_gpu(ctx, small, "bgra", ow, oh), exp_small, "cfg5 lanczos")


def test_opacity_zero_layer_is_identity_and_constant_stays_constant(ctx):
    w, h = 1920, 1080
    canvas0 = util.alloc_image("nv12", w, h, seed=5)
    src = util.alloc_image("bgra", 640, 360, seed=6)
    gd = G.to_gpu(ctx, "nv12", w, h, canvas0)
    gs = G.to_gpu(ctx, "bgra", 640, 360, src)
    u = util.make_uniforms((w, h), rect=(100, 100, 1280, 720), opacity=0.0, in_size=(640, 360))
    sv.usingContext(ctx, lambda c: sv.runComputeKernel(c, images=[gs], target=gd, kernel=sv.ComputeKernel.img_bgra_nv12,
                                                       uniforms=u, blends=True))
    G.assert_same(G.from_gpu(ctx, gd, "nv12", w, h), canvas0, "opacity 0")
    # a constant NV12 picture stays constant through bilinear scaling + integer conversion
    const = [np.full((1080, 1920), 120, np.uint8), np.full((540, 960, 2), 128, np.uint8)]
    const[1][..., 0] = 90
    const[1][..., 1] = 200
    gc = G.to_gpu(ctx, "nv12", 1920, 1080, const)
    out = G.to_gpu(ctx, "bgra", 1280, 720, util.alloc_image("bgra", 1280, 720))
    sv.usingContext(ctx, lambda c: sv.compositeTick(c, out, [(sv.ComputeKernel.img_nv12_bgra, gc,
                                                                util.full_canvas_uniforms((1280, 720), (1920, 1080)), 0)], True))
    got = G.from_gpu(ctx, out, "bgra", 1280, 720)[0]
    r, g, b = O.yuv2rgb_int(0, 120, 90, 200)
    assert np.all(got == np.array([b, g, r, 255], dtype=np.uint8))


def test_pipeline_with_a_rotated_logo_full_size(ctx):
    """The headline tick plus one rotated RGBA logo at full size: the rotated layer is applied per pixel inside tick_bgra_wave, the
    video layers stay on the staged path; a batch of 12 ticks (enough strips for 16-row selection logic to be exercised either way)."""
    sw, sh, dw, dh = 1920, 1080, 1280, 720
    srcs = [util.alloc_image("nv12", sw, sh, seed=0x5EED0000 + 300 + i) for i in range(4)]
    us = [util.full_canvas_uniforms((dw, dh), (sw, sh), opacity=o) for o in (1.0, 0.75, 0.5, 0.25)]
    logo = util.alloc_image("rgba", 320, 180, seed=0x5EED0000 + 310)
    lu = util.make_uniforms((dw, dh), rect=(820, 60, 320, 180), rotation=0.3, opacity=0.9, in_size=(320, 180))
    exp = util.alloc_image("bgra", dw, dh)
    assert O.run_kernel("img_clear_bgra", exp, threads=CORES) == 0
    for s, u in zip(srcs, us):
        assert O.run_kernel("img_nv12_bgra", exp, s, u, threads=CORES) == 0
    assert O.run_kernel("img_rgba_bgra_tx", exp, logo, lu, threads=CORES) == 0
    gs = [G.to_gpu(ctx, "nv12", sw, sh, s) for s in srcs]
    gl = G.to_gpu(ctx, "rgba", 320, 180, logo)
    K = sv.ComputeKernel
    layers = [(K.img_nv12_bgra, g, u, 0) for g, u in zip(gs, us)] + [(K.img_rgba_bgra_tx, gl, lu, 0)]
    gds = [G.to_gpu(ctx, "bgra", dw, dh, util.alloc_image("bgra", dw, dh, seed=7 + i)) for i in range(12)]
    h, name, keep = G.make_batch(ctx, [(gd, True, layers) for gd in gds])
    assert name == "tick_bgra_stream + tick_bgra_wave", name      # (the four videos, then the logo in the strips it touches)
    G.run_batch(ctx, h)
    G.destroy_batch(h)
    for i in (0, 5, 11):
        G.assert_same(G.from_gpu(ctx, gds[i], "bgra", dw, dh), exp, f"tick {i}")


def test_large_launch_of_planar_layers_takes_the_wave_instantiation(ctx):
    """cfg2 with a y420p source: small launches use tick_y420p_bgra_tiled, launches of >= 8192 strips the y420p-only instantiation of
    tick_bgra_wave (the faster one there) — same bytes either way."""
    sw, sh, dw, dh = 1920, 1080, 1280, 720
    src = util.alloc_image("y420p", sw, sh, seed=0x5EED0000 + 320)
    u = util.full_canvas_uniforms((dw, dh), (sw, sh))
    exp = util.alloc_image("bgra", dw, dh)
    assert O.run_kernel("img_clear_bgra", exp, threads=CORES) == 0
    assert O.run_kernel("img_y420p_bgra", exp, src, u, threads=CORES) == 0
    gs = G.to_gpu(ctx, "y420p", sw, sh, src)
    K = sv.ComputeKernel
    for n, want in ((2, "tick_y420p_bgra_tiled"), (12, "tick_bgra_wave")):
        gds = [G.to_gpu(ctx, "bgra", dw, dh, util.alloc_image("bgra", dw, dh, seed=9 + i)) for i in range(n)]
        h, name, keep = G.make_batch(ctx, [(gd, True, [(K.img_y420p_bgra, gs, u, 0)]) for gd in gds])
        assert name == want, (n, name)
        G.run_batch(ctx, h)
        G.destroy_batch(h)
        for gd in (gds[0], gds[-1]):
            G.assert_same(G.from_gpu(ctx, gd, "bgra", dw, dh), exp, f"{n} ticks via {name}")
