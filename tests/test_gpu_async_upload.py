"""Asynchronous uploads from caller-pinned memory (chv_host_alloc + chv_upload(async=2)) on a side context, ordered
against kernels of another context by the per-buffer upload events and against reuse of the pinned frame by
chv_event_wait — the hipMemcpyAsync-on-a-side-stream arrangement of north_star / SURVEY section 8d (cfg4 end-to-end),
and the stream timers (chv_event_*)."""
import ctypes as C

import numpy as np
import pytest

import gpuutil as G
import util
from oracle import oracle as O
from swiftvideo_amd import chipvideo as cv
from swiftvideo_amd import compute as sv

pytestmark = pytest.mark.gpu


def test_pinned_uploads_overlap_with_kernels_and_stay_ordered(ctx):
    lib = cv.load()
    up = sv.createComputeContext(sharing=ctx)            # GPUBarrierUpload's own context (compute.swift:177)
    (W, H), (w, h) = (320, 180), (160, 90)
    ysz, csz = W * H, W * H // 2
    n_frames, ring = 12, 3                               # more frames than pinned slots: slots are recycled
    pinned = C.c_void_p()
    cv.check(lib.chv_host_alloc(up.handle, ring * (ysz + csz), C.byref(pinned)))
    host = np.ctypeslib.as_array(C.cast(pinned, C.POINTER(C.c_uint8)), shape=(ring * (ysz + csz),))
    # one device picture per pinned slot; uploads overwrite them, kernels read them
    dev = [G.to_gpu(ctx, "nv12", W, H, util.alloc_image("nv12", W, H)) for _ in range(ring)]
    outs = [G.to_gpu(ctx, "bgra", w, h, util.alloc_image("bgra", w, h)) for _ in range(n_frames)]
    u = util.full_canvas_uniforms((w, h), (W, H))
    done = [C.c_void_p() for _ in range(ring)]           # "the kernel that read slot k has run"
    for e in done:
        cv.check(lib.chv_event_create(ctx.handle, C.byref(e)))
        cv.check(lib.chv_event_record(ctx.handle, e))
    t0, t1 = C.c_void_p(), C.c_void_p()
    cv.check(lib.chv_event_create(ctx.handle, C.byref(t0)))
    cv.check(lib.chv_event_create(ctx.handle, C.byref(t1)))
    cv.check(lib.chv_event_record(ctx.handle, t0))
    expected = []
    for f in range(n_frames):
        k = f % ring
        src = util.alloc_image("nv12", W, H, seed=300 + f)
        exp = util.alloc_image("bgra", w, h)
        assert O.run_kernel("img_clear_bgra", exp) == 0 and O.run_kernel("img_nv12_bgra", exp, src, u) == 0
        expected.append(exp)
        # the pinned slot may only be rewritten once the previous upload from it has been passed by `up`'s stream,
        # and the device picture only once the kernel that read it has run
        cv.check(lib.chv_event_wait(up.handle, done[k]))
        sv.endComputePass(up, True)
        base = k * (ysz + csz)
        host[base:base + ysz] = src[0].reshape(-1)
        host[base + ysz:base + ysz + csz] = src[1].reshape(-1)
        img = dev[k].imageBuffer()
        tex, pitch, off = img.computeTextures, img.gpuPitches, img.gpuOffsets
        if f & 1:      # plane by plane ...
            cv.check(lib.chv_upload(up.handle, tex[0]._h, off[0], pitch[0], host[base:].ctypes.data, W, W, H, 2))
            cv.check(lib.chv_upload(up.handle, tex[1]._h, off[1], pitch[1], host[base + ysz:].ctypes.data, W, W, H // 2, 2))
        else:          # ... or the whole NV12 picture as one pitched region (the planes are adjacent on both sides)
            assert tex[0] is tex[1] and pitch[0] == pitch[1] and off[1] == off[0] + pitch[0] * H
            cv.check(lib.chv_upload(up.handle, tex[0]._h, off[0], pitch[0], host[base:].ctypes.data, W, W, H + H // 2, 2))
        # no host-side wait: the kernel's stream waits for the two uploads through the buffers' events
        layer = (sv.ComputeKernel.img_nv12_bgra, dev[k], u, 0)
        sv.compositeTick(ctx, outs[f], [layer], clearFirst=True)
        cv.check(lib.chv_event_record(ctx.handle, done[k]))
    cv.check(lib.chv_event_record(ctx.handle, t1))
    cv.check(lib.chv_event_synchronize(t1))
    ms = C.c_float(-1)
    cv.check(lib.chv_event_elapsed_ms(t0, t1, C.byref(ms)))
    assert 0.0 < ms.value < 5000.0
    sv.endComputePass(ctx, True)
    sv.endComputePass(up, True)
    for f in range(n_frames):
        G.assert_same(G.from_gpu(ctx, outs[f], "bgra", w, h), expected[f], f"frame {f}")
    for e in done + [t0, t1]:
        cv.check(lib.chv_event_destroy(e))
    cv.check(lib.chv_host_free(up.handle, pinned))
    sv.destroyComputeContext(up)
    # argument checks
    assert lib.chv_event_elapsed_ms(None, None, C.byref(ms)) != 0
    assert lib.chv_host_alloc(ctx.handle, 0, C.byref(pinned)) != 0


def test_async_downloads_overlap_with_kernels_and_stay_ordered(ctx):
    """chv_download_async on a download context of its own (GPUBarrierDownload's, compute.swift:217-255): tick t's canvas is read back into a
    pinned ring while tick t + 1 is composed; order comes from events only (kernel -> copy, copy -> reuse of the canvas and of the pinned slot),
    the host waits once at the end — and every frame in the ring is the oracle's"""
    lib = cv.load()
    dl = sv.createComputeContext(sharing=ctx)
    (W, H), (w, h) = (256, 88), (256, 88)                # (PictureSlab: packed rows of a multiple of 128 bytes)
    n_ticks, ring = 10, 3
    slab = sv.PictureSlab(ctx, (w, h), sv.PixelFormat.nv12, ring)           # the canvases: NV12 frames, one allocation
    fb = slab.frameBytes
    pinned = C.c_void_p()
    cv.check(lib.chv_host_alloc(dl.handle, n_ticks * fb, C.byref(pinned)))
    host = np.ctypeslib.as_array(C.cast(pinned, C.POINTER(C.c_uint8)), shape=(n_ticks * fb,))
    host[:] = 0xA5
    u = util.full_canvas_uniforms((w, h), (W, H))
    k = sv.defaultComputeKernelFromString("img_bgra_nv12_int")
    composed = [C.c_void_p() for _ in range(ring)]       # "the tick that wrote canvas k has run"
    copied = [C.c_void_p() for _ in range(ring)]         # "the copy that read canvas k has run"
    for e in composed + copied:
        cv.check(lib.chv_event_create(ctx.handle, C.byref(e)))
    for e in copied:
        cv.check(lib.chv_event_record(dl.handle, e))
    expected, keep = [], []
    for t in range(n_ticks):
        c = t % ring
        src = util.alloc_image("bgra", W, H, seed=500 + t)
        exp = util.alloc_image("nv12", w, h)
        assert O.run_kernel("img_clear_nv12", exp) == 0 and O.run_kernel("img_bgra_nv12_int", exp, src, u) == 0
        expected.append(exp)
        g = G.to_gpu(ctx, "bgra", W, H, src)
        keep.append(g)
        cv.check(lib.chv_event_wait(ctx.handle, copied[c]))                  # the canvas is free again
        sv.beginComputePass(ctx)
        sv.compositeTick(ctx, slab.pictures[c], [(k, g, u, 0)], True)
        sv.endComputePass(ctx, False)                                        # no host wait
        cv.check(lib.chv_event_record(ctx.handle, composed[c]))
        cv.check(lib.chv_event_wait(dl.handle, composed[c]))
        slab.download(dl, c, 1, pinned.value + t * fb)
        cv.check(lib.chv_event_record(dl.handle, copied[c]))
    sv.endComputePass(dl, True)                                              # the one host wait
    for t, exp in enumerate(expected):
        got = host[t * fb:(t + 1) * fb]
        assert np.array_equal(got[:w * h].reshape(h, w), exp[0]), f"tick {t}: luma"
        assert np.array_equal(got[w * h:].reshape(h // 2, w // 2, 2), exp[1]), f"tick {t}: chroma"
    for e in composed + copied:
        cv.check(lib.chv_event_destroy(e))
    cv.check(lib.chv_host_free(dl.handle, pinned))
    sv.destroyComputeContext(dl)


def test_async_download_argument_checks(ctx):
    """chv_download_async fails like chv_download does (same span checks: the reference's downloadComputeBuffer throws on a size mismatch,
    compute.cl.swift:381-396) and leaves the context usable"""
    lib = cv.load()
    h = C.c_void_p()
    cv.check(lib.chv_buffer_alloc(ctx.handle, 1024, C.byref(h)))
    pinned = C.c_void_p()
    cv.check(lib.chv_host_alloc(ctx.handle, 4096, C.byref(pinned)))
    ok = lambda *a: lib.chv_download_async(*a)
    assert ok(None, pinned, 64, h, 0, 64, 64, 16) != 0                  # no context
    assert ok(ctx.handle, None, 64, h, 0, 64, 64, 16) != 0              # no destination
    assert ok(ctx.handle, pinned, 64, None, 0, 64, 64, 16) != 0         # no source
    assert ok(ctx.handle, pinned, 64, h, 0, 64, 64, 17) != 0            # 17 rows of 64 bytes leave the 1024-byte buffer
    assert ok(ctx.handle, pinned, 64, h, 1000, 64, 64, 1) != 0          # offset + row beyond the end
    assert ok(ctx.handle, pinned, 32, h, 0, 64, 64, 16) != 0            # destination pitch smaller than a row
    assert ok(ctx.handle, pinned, 64, h, 0, 64, 64, 16) == 0            # the whole buffer
    sv.endComputePass(ctx, True)
    cv.check(lib.chv_host_free(ctx.handle, pinned))
    cv.check(lib.chv_buffer_free(h))
