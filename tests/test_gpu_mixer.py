"""VideoMixer semantics (mix.video.swift:95-165): backing ring of 10, z-order, carry-over of the last
tick's samples, fused and reference-sequence ticks give the same bytes, errors surface as events."""
import numpy as np
import pytest

import gpuutil as G
import util
from oracle import oracle as O
from swiftvideo_amd import compute as sv

pytestmark = pytest.mark.gpu


def _layer(ctx, fmt, w, h, seed, canvas, rect, z, opacity=1.0, asset="cam"):
    M = util.ortho(*canvas) @ util._mat_translate(rect[0], rect[1]) @ util._mat_scale(rect[2], rect[3])
    p = sv.pictureFromArrays(G.FMT[fmt], (w, h), util.alloc_image(fmt, w, h, seed=seed), matrix=M, opacity=opacity,
                             zIndex=z, assetId=asset)
    return sv.uploadComputePicture(ctx, p)


@pytest.mark.parametrize("fmt", ["nv12", "y420p"])
def test_mixer_tick_matches_oracle_in_z_order(ctx, fmt):
    canvas = (96, 54)
    mixer = sv.VideoMixer("ws", 1 / 30, canvas, outputFormat=G.FMT[fmt], computeContext=ctx, fused=True)
    seq = sv.VideoMixer("ws", 1 / 30, canvas, outputFormat=G.FMT[fmt], computeContext=ctx, fused=False)
    specs = [("bgra", 40, 30, 11, (30, 10, 50, 30), 2, 0.8), (fmt, 64, 36, 12, (0, 0, 96, 54), 0, 1.0),
             ("rgba", 20, 20, 13, (5, 5, 30, 30), 1, 0.6)]
    for m in (mixer, seq):
        for s, w, h, seed, rect, z, op in specs:            # pushed out of z order on purpose
            assert m.push(_layer(ctx, s, w, h, seed, canvas, rect, z, op))[0] == "nothing"
    exp = util.alloc_image(fmt, *canvas)
    assert O.run_kernel(f"img_clear_{fmt}", exp) == 0
    for s, w, h, seed, rect, z, op in sorted(specs, key=lambda t: t[5]):
        u = util.make_uniforms(canvas, rect=rect, opacity=op, in_size=(w, h))
        assert O.run_kernel(f"img_{s}_{fmt}", exp, util.alloc_image(s, w, h, seed=seed), u) == 0
    out, out2 = mixer.mix(at=1.0), seq.mix(at=1.0)
    assert out is not None and out2 is not None, (mixer.result, seq.result)
    G.assert_same(G.from_gpu(ctx, out, fmt, *canvas), exp, "fused tick")
    G.assert_same(G.from_gpu(ctx, out2, fmt, *canvas), exp, "sequential tick")
    # next tick without new samples re-uses last tick's (samples[1]); the one after that is just the clear
    out3 = mixer.mix(at=2.0)
    G.assert_same(G.from_gpu(ctx, out3, fmt, *canvas), exp, "carry-over tick")
    out4 = mixer.mix(at=3.0)
    clr = util.alloc_image(fmt, *canvas)
    assert O.run_kernel(f"img_clear_{fmt}", clr) == 0
    G.assert_same(G.from_gpu(ctx, out4, fmt, *canvas), clr, "empty tick")


def test_backing_ring_has_ten_images(ctx):
    mixer = sv.VideoMixer("ws", 1 / 30, (32, 18), outputFormat=sv.PixelFormat.nv12, computeContext=ctx)
    seen = []
    for t in range(23):
        out = mixer.mix(at=float(t))
        seen.append(id(out.imageBuffer().computeTextures[0]))
    assert len(set(seen)) == 10                       # numberBackingImages, mix.video.swift:167
    assert seen[10:20] == seen[0:10] and seen[20:23] == seen[0:3]


def test_sample_from_own_asset_passes_through_and_errors_become_events(ctx):
    mixer = sv.VideoMixer("ws", 1 / 30, (32, 18), outputFormat=sv.PixelFormat.y420p, computeContext=ctx, assetId="mixer")
    own = sv.createPictureSample((32, 18), sv.PixelFormat.y420p, assetId="mixer")
    assert mixer.push(own) == ("just", own)           # mix.video.swift:66-73
    # an NV12 layer onto a y420p canvas: findKernel synthesises img_nv12_y420p -> invalidValue -> event error
    bad = _layer(ctx, "nv12", 16, 8, 3, (32, 18), (0, 0, 32, 18), 0)
    mixer.push(bad)
    assert mixer.mix(at=0.0) is None
    assert mixer.result[0] == "error" and mixer.result[1][0] == "mix.video" and mixer.result[1][1] == -2
    # the mixer keeps working on the next tick once the bad sample has aged out
    mixer.mix(at=1.0)
    assert mixer.mix(at=2.0) is not None


@pytest.mark.parametrize("src_fmt,dst_fmt,kernel", [("nv12", "bgra", "img_nv12_bgra"), ("y420p", "bgra", "img_y420p_bgra"),
                                                    ("bgra", "nv12", "img_bgra_nv12_int"), ("rgba", "y420p", "img_rgba_y420p_int"),
                                                    ("bgra", "nv12", "img_bgra_nv12"), ("rgba", "y420p", "img_rgba_y420p"),
                                                    ("y420p", "nv12", "img_y420p_nv12"), ("bgra", "bgra", "img_bgra_bgra_tx")])
def test_picture_filter_converts_and_scales(ctx, src_fmt, dst_fmt, kernel):
    """PictureFilter (the operator filter.pict.swift:20-47 sketches): one full-canvas layer = convert + bilinear
    scale; equals the oracle's kernel run with the literal full-canvas uniforms after a clear."""
    (W, H), (w, h) = (192, 108), (128, 72)
    planes = util.alloc_image(src_fmt, W, H, seed=91)
    cpu = sv.pictureFromArrays(G.FMT[src_fmt], (W, H), planes, assetId="cam", time=3.0, pts=3.5, zIndex=7)
    # RGB -> 4:2:0: the integer BT.601/709 matrix by default, the reference's float kernels with integerMatrix=False
    filt = sv.PictureFilter((w, h), G.FMT[dst_fmt], computeContext=ctx, integerMatrix=kernel.endswith("_int"))
    tag, out = filt(cpu)                                   # CPU sample: uploaded by the filter
    assert tag == "just", out
    assert out.bufferType() == "gpu" and out.size() == (w, h) and out.pixelFormat() == G.FMT[dst_fmt]
    assert (out.assetId(), out.time(), out.pts(), out.zIndex()) == ("cam", 3.0, 3.5, 7)
    exp = util.alloc_image(dst_fmt, w, h)
    assert O.run_kernel(f"img_clear_{dst_fmt}", exp) == 0
    assert O.run_kernel(kernel, exp, planes, util.full_canvas_uniforms((w, h), (W, H))) == 0
    G.assert_same(G.from_gpu(ctx, out, dst_fmt, w, h), exp, kernel)
    tag2, out2 = filt(sv.uploadComputePicture(ctx, cpu))   # GPU sample: used in place, next image of the ring
    assert tag2 == "just" and out2.imageBuffer().computeTextures[0] is not out.imageBuffer().computeTextures[0]
    G.assert_same(G.from_gpu(ctx, out2, dst_fmt, w, h), exp, kernel + " (gpu input)")


def test_picture_filter_lanczos_and_errors(ctx):
    (W, H), (w, h) = (160, 90), (64, 36)
    planes = util.alloc_image("bgra", W, H, seed=92)
    pic = sv.pictureFromArrays(sv.PixelFormat.BGRA, (W, H), planes)
    tag, out = sv.PictureFilter((w, h), sv.PixelFormat.BGRA, computeContext=ctx, scaler="lanczos")(pic)
    assert tag == "just"
    exp = util.alloc_image("bgra", w, h)
    assert O.lanczos_bgra(exp[0], planes[0]) == 0
    G.assert_same(G.from_gpu(ctx, out, "bgra", w, h), exp, "lanczos filter")
    # no kernel for this pair (nv12 -> rgba is in no table): surfaces as an error event, the filter stays usable
    nv = sv.pictureFromArrays(sv.PixelFormat.nv12, (W, H), util.alloc_image("nv12", W, H, seed=93))
    f = sv.PictureFilter((w, h), sv.PixelFormat.RGBA, computeContext=ctx)
    tag, err = f(nv)
    assert tag == "error" and err[0] == "filter.pict"
    tag, err = sv.PictureFilter((w, h), sv.PixelFormat.BGRA, computeContext=ctx, scaler="lanczos")(nv)
    assert tag == "error"
    with pytest.raises(sv.ComputeError):
        sv.PictureFilter((w, h), scaler="bicubic")


def test_mixer_group_ticks_many_mixers_in_one_launch(ctx):
    """VideoMixerGroup / TickBatch: N mixers of a device composed by one launch give the bytes each mixer's own
    mix() gives (chv_batch_* vs chv_composite)."""
    specs = [("nv12", (96, 54)), ("y420p", (64, 36)), ("bgra", (80, 44)), ("bgra", (80, 44))]
    group_mixers, solo_mixers = [], []
    for k, (fmt, canvas) in enumerate(specs):
        for dest in (group_mixers, solo_mixers):
            m = sv.VideoMixer("ws", 1 / 30, canvas, outputFormat=G.FMT[fmt], computeContext=ctx, assetId=f"mixer{k}")
            src_fmt = "nv12" if fmt != "y420p" else "y420p"
            m.push(_layer(ctx, src_fmt, 48, 30, 40 + k, canvas, (0, 0) + canvas, 0))
            m.push(_layer(ctx, "bgra", 20, 16, 50 + k, canvas, (8, 6, 30, 20), 1, opacity=0.7, asset="logo"))
            dest.append(m)
    group = sv.VideoMixerGroup(group_mixers)
    outs = group.mix(at=1.0)
    assert outs is not None and len(outs) == len(specs), [m.result for m in group_mixers]
    for k, ((fmt, canvas), out, solo) in enumerate(zip(specs, outs, solo_mixers)):
        ref = solo.mix(at=1.0)
        assert out.assetId() == f"mixer{k}" and out.time() == 1.0
        G.assert_same(G.from_gpu(ctx, out, fmt, *canvas), G.from_gpu(ctx, ref, fmt, *canvas), f"mixer {k} ({fmt})")
    # a TickBatch on its own: two clears of same-format canvases in one launch; mixed formats are refused
    with pytest.raises(sv.ComputeError):
        sv.TickBatch(ctx, [(outs[0], True, []), (outs[2], True, [])])
    b = sv.TickBatch(ctx, [(outs[2], True, []), (outs[3], True, [])])
    assert b.count == 2 and b.launches >= 1 and b.kernelName
    sv.usingContext(ctx, b.run)
    b.destroy()
