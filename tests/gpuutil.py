"""Helpers for the -m gpu parity tests: run the same job through the HIP path (C ABI via
swiftvideo_amd.compute) and through the oracle, on identical seeded inputs."""
import numpy as np

import util
from oracle import oracle as O
from swiftvideo_amd import compute as sv

FMT = {"nv12": sv.PixelFormat.nv12, "y420p": sv.PixelFormat.y420p, "bgra": sv.PixelFormat.BGRA,
       "rgba": sv.PixelFormat.RGBA}


def kernel_formats(name):
    """img_<src>_<dst>[_tx] -> (src fmt, dst fmt)"""
    parts = name.split("_")
    return parts[1], parts[2]


def to_gpu(ctx, fmt, w, h, planes, **kw):
    pict = sv.pictureFromArrays(FMT[fmt], (w, h), planes, **kw)
    return sv.uploadComputePicture(ctx, pict)


def from_gpu(ctx, sample, fmt, w, h):
    """Download and return plane arrays shaped like util.alloc_image's."""
    cpu = sv.downloadComputePicture(ctx, sample, retainGpuBuffer=True)
    out = []
    for buf, (r, c, comps) in zip(cpu.imageBuffer().buffers, util.plane_shapes(fmt, w, h)):
        r, c = max(r, 1), max(c, 1)
        a = buf[:r, : c * comps]
        out.append(a if comps == 1 else a.reshape(r, c, comps))
    return out


def assert_same(got, exp, what=""):
    for i, (g, e) in enumerate(zip(got, exp)):
        if not np.array_equal(g, e):
            diff = np.argwhere(g != e)
            raise AssertionError(f"{what}: plane {i} differs at {len(diff)} positions, first {diff[0].tolist()}: "
                                 f"hip={g[tuple(diff[0])]} oracle={e[tuple(diff[0])]}, "
                                 f"max |d|={np.abs(g.astype(int) - e.astype(int)).max()}")


def run_both(ctx, kernel, cw, ch, iw, ih, uniforms, seed, csc=0, clear_first=False):
    """One layer onto a seeded canvas through chv_run_kernel (or a cleared canvas through
    chv_composite); returns (hip planes, oracle planes)."""
    s, d = kernel_formats(kernel)
    src = util.alloc_image(s, iw, ih, seed=seed)
    canvas0 = util.alloc_image(d, cw, ch, seed=seed + 1000)
    exp = util.copy_image(canvas0)
    if clear_first:
        assert O.run_kernel(f"img_clear_{d}", exp) == 0
    assert O.run_kernel(kernel, exp, src, uniforms, csc=csc) == 0
    gsrc = to_gpu(ctx, s, iw, ih, src)
    gdst = to_gpu(ctx, d, cw, ch, canvas0)
    k = sv.defaultComputeKernelFromString(kernel)
    if clear_first:
        sv.usingContext(ctx, lambda c: sv.compositeTick(c, gdst, [(k, gsrc, uniforms, csc)], True))
    else:
        sv.usingContext(ctx, lambda c: sv.runComputeKernel(c, images=[gsrc], target=gdst, kernel=k,
                                                           uniforms=uniforms, blends=True, colorspace=csc))
    return from_gpu(ctx, gdst, d, cw, ch), exp


def make_batch(ctx, ticks):
    """ticks: [(target PictureSample, clear_first, [(kernel, sample, uniforms, csc)])] -> (handle, kernel name, keepalive).  The keepalive holds the
    descriptor arrays AND the tick list itself: a batch borrows its pictures (device pointers), and a picture created inline in the argument
    would otherwise be freed the moment this returns."""
    import ctypes as C
    from swiftvideo_amd import chipvideo as cv
    lib = cv.load()
    arr = (cv.Tick * len(ticks))()
    keep = []
    for i, (target, clear, layers) in enumerate(ticks):
        la = sv._layer_array(layers)
        keep.append(la)
        arr[i].target = sv._image_desc(target)
        arr[i].clear_first = 1 if clear else 0
        arr[i].n_layers = len(layers)
        arr[i].layers = la
    h = C.c_void_p()
    cv.check(lib.chv_batch_create(ctx.handle, arr, len(ticks), C.byref(h)))
    name = C.create_string_buffer(128)
    cv.check(lib.chv_batch_describe(h, name, 128, None))
    return h, name.value.decode(), (arr, keep, ticks)


def run_batch(ctx, h):
    from swiftvideo_amd import chipvideo as cv
    lib = cv.load()
    cv.check(lib.chv_pass_begin(ctx.handle))
    cv.check(lib.chv_batch_run(ctx.handle, h))
    cv.check(lib.chv_pass_end(ctx.handle, 1))


def destroy_batch(h):
    from swiftvideo_amd import chipvideo as cv
    cv.check(cv.load().chv_batch_destroy(h))
