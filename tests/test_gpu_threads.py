"""Threading contract of the boundary: one thread per context at a time, any number of
contexts per device entered concurrently (the mixer's serial queue and the upload/download barriers' bus queues each
own a context created with createComputeContext(sharing:), compute.swift:177,234; mix.video.swift:55,99), and
ComputeBuffer deinit — chv_buffer_free — from whatever thread drops the last reference."""
import threading

import numpy as np
import pytest

import gpuutil as G
import util
from oracle import oracle as O
from swiftvideo_amd import compute as sv

pytestmark = pytest.mark.gpu


def test_concurrent_contexts_and_cross_thread_free(ctx):
    n_threads, n_iter = 6, 12
    (W, H), (w, h) = (160, 90), (96, 54)
    errors, garbage = [], []
    lock = threading.Lock()

    def worker(k):
        try:
            c = sv.createComputeContext(sharing=ctx)
            fmt = ("nv12", "y420p", "bgra")[k % 3]
            kernel = {"nv12": "img_nv12_nv12", "y420p": "img_y420p_y420p", "bgra": "img_nv12_bgra"}[fmt]
            src_fmt = "y420p" if fmt == "y420p" else "nv12"
            u = util.full_canvas_uniforms((w, h), (W, H), opacity=0.75)
            for it in range(n_iter):
                src = util.alloc_image(src_fmt, W, H, seed=1000 * k + it)
                exp = util.alloc_image(fmt, w, h)
                assert O.run_kernel(f"img_clear_{fmt}", exp) == 0
                assert O.run_kernel(kernel, exp, src, u) == 0
                # uploads go through this thread's own context; asynchronous ones exercise the per-buffer events
                gs = sv.uploadComputePicture(c, sv.pictureFromArrays(G.FMT[src_fmt], (W, H), src), asynchronous=bool(it & 1))
                gd = G.to_gpu(c, fmt, w, h, util.alloc_image(fmt, w, h))
                layer = (sv.defaultComputeKernelFromString(kernel), gs, u, 0)
                sv.usingContext(c, lambda cc: sv.compositeTick(cc, gd, [layer], clearFirst=True))
                got = G.from_gpu(c, gd, fmt, w, h)
                for a, b in zip(got, exp):
                    if not np.array_equal(a, b):
                        raise AssertionError(f"thread {k} iteration {it}: {kernel} differs from the oracle")
                with lock:
                    garbage.append((gs, gd))           # dropped by the main thread: free on a foreign thread
            sv.destroyComputeContext(c)
        except Exception as e:                          # noqa: BLE001 - report through the main thread
            with lock:
                errors.append(f"{type(e).__name__}: {e}")

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(n_threads)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert len(garbage) == n_threads * n_iter
    garbage.clear()                                     # ComputeBuffer.__del__ -> chv_buffer_free, on this thread
    # the parent context is still healthy
    src = util.alloc_image("nv12", W, H, seed=5)
    exp = util.alloc_image("bgra", w, h)
    u = util.full_canvas_uniforms((w, h), (W, H))
    assert O.run_kernel("img_clear_bgra", exp) == 0 and O.run_kernel("img_nv12_bgra", exp, src, u) == 0
    gd = G.to_gpu(ctx, "bgra", w, h, util.alloc_image("bgra", w, h))
    layer = (sv.ComputeKernel.img_nv12_bgra, G.to_gpu(ctx, "nv12", W, H, src), u, 0)
    sv.usingContext(ctx, lambda cc: sv.compositeTick(cc, gd, [layer], clearFirst=True))
    G.assert_same(G.from_gpu(ctx, gd, "bgra", w, h), exp, "after the threads")
