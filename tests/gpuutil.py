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
