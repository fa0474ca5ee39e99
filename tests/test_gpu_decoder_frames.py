"""Decoder adjacency: pictures laid out the way FFmpeg hands them over — `AVFrame.data[i]` per plane, each
with its own `linesize[i]` larger than the row (dec.video.ffmpeg.swift:143-184: one `Data` per plane, `Plane.stride =
linesize[idx]`, chroma planes half size for YUV420P / NV12) — go through GPUBarrierUpload and the mixer exactly like tightly
packed pictures: the padding bytes never reach a pixel.  No FFmpeg in the image: the frames are built here with FFmpeg's
layout rules (linesize = width rounded up to 64 and then some, planes in separate allocations)."""
import numpy as np
import pytest

import gpuutil as G
import util
from oracle import oracle as O
from swiftvideo_amd import compute as sv

pytestmark = pytest.mark.gpu

PF = {"nv12": sv.PixelFormat.nv12, "y420p": sv.PixelFormat.y420p, "bgra": sv.PixelFormat.BGRA}


def av_frame(fmt, w, h, seed, extra=0):
    """(tight planes for the oracle, [data[i]] arrays of shape (rows, linesize[i]) with garbage in the padding)"""
    tight = util.alloc_image(fmt, w, h, seed=seed)
    data = []
    for k, p in enumerate(tight):
        rows = p.shape[0]
        row_bytes = p.shape[1] * (1 if p.ndim == 2 else p.shape[2])
        linesize = (row_bytes + 63) // 64 * 64 + extra
        buf = util.splitmix_bytes(seed * 31 + k, rows * linesize).reshape(rows, linesize).copy()    # padding: anything but zeros
        buf[:, :row_bytes] = p.reshape(rows, row_bytes)
        data.append(buf)
    return tight, data


@pytest.mark.parametrize("fmt,w,h,extra", [("y420p", 854, 480, 0), ("y420p", 1280, 720, 64), ("nv12", 854, 480, 32), ("nv12", 1920, 1080, 0),
                                            ("y420p", 176, 144, 0), ("bgra", 300, 200, 0)])
@pytest.mark.parametrize("own_canvas", [True, False])
def test_av_frame_layout_through_upload_and_mixer(ctx, fmt, w, h, extra, own_canvas):
    # the reference's kernel table has no NV12 -> y420p entry (findKernel would fail): every source on the 4:2:0 canvas it has a
    # kernel for, and on a BGRA canvas
    canvas = {"y420p": "y420p", "nv12": "nv12", "bgra": "y420p"}[fmt] if own_canvas else "bgra"
    cw, ch = 640, 360
    tight, data = av_frame(fmt, w, h, seed=901 + w, extra=extra)
    assert all(d.shape[1] > t.shape[1] * (1 if t.ndim == 2 else t.shape[2]) or extra == 0 for d, t in zip(data, tight))
    pic = sv.pictureFromArrays(PF[fmt], (w, h), data, matrix=util.ortho(cw, ch) @ util._mat_scale(cw, ch), zIndex=0, assetId="dec")
    assert [p.stride for p in pic.imageBuffer().planes] == [d.shape[1] for d in data]          # Plane.stride = linesize
    up = sv.GPUBarrierUpload(ctx)
    tag, gpu = up(pic)
    assert tag == "just"
    family = {} if fmt != "bgra" or canvas != "bgra" else {"bgraKernelFamily": "tx"}
    mixer = sv.VideoMixer("ws", 1 / 30, (cw, ch), outputFormat=PF[canvas], computeContext=ctx, fused=True, **family)
    assert mixer.push(gpu)[0] == "nothing"
    out = mixer.mix(at=0.0)
    assert out is not None, mixer.result
    exp = util.alloc_image(canvas, cw, ch)
    assert O.run_kernel(f"img_clear_{canvas}", exp) == 0
    kernel = f"img_{fmt}_{canvas}" + ("_tx" if family else "")
    assert O.run_kernel(kernel, exp, tight, util.full_canvas_uniforms((cw, ch), (w, h)), threads=4) == 0
    G.assert_same(G.from_gpu(ctx, out, canvas, cw, ch), exp, f"{fmt} {w}x{h} linesize {[d.shape[1] for d in data]} -> {canvas}")


def test_av_frame_round_trip_keeps_only_the_pixels(ctx):
    """upload + download of a padded frame returns the pixels; the download is tightly pitched or padded as the library
    chooses, but never carries the source's padding bytes into the picture area"""
    w, h = 854, 480
    tight, data = av_frame("y420p", w, h, seed=77, extra=64)
    pic = sv.pictureFromArrays(sv.PixelFormat.y420p, (w, h), data)
    gpu = sv.uploadComputePicture(ctx, pic)
    back = sv.downloadComputePicture(ctx, gpu).imageBuffer()
    for t, b, p in zip(tight, back.buffers, back.planes):
        assert np.array_equal(np.asarray(b).reshape(t.shape[0], -1)[:, :t.shape[1]], t)
