"""`ComputeKernel.custom(name:)` + buildComputeKernel (compute.swift:72-73, compute.cl.swift:153-232) through
hipRTC: a user kernel sees the target planes, the current planes when `blends`, the input images and the
uniform bytes, in the reference's binding order."""
import numpy as np
import pytest

import gpuutil as G
import util
from oracle import oracle as O
from swiftvideo_amd import chipvideo as cv
from swiftvideo_amd import compute as sv

pytestmark = pytest.mark.gpu

INVERT = r'''
// out = 1 - in on B,G,R (nearest, same size), alpha from the CURRENT target when blends, else 1
extern "C" __global__ void invert_bgra(chv_custom_args a) {
    const chv_dev_plane &dst = a.target.planes[0];
    CHV_GUARD(dst);
    const int x = CHV_GID_X, y = CHV_GID_Y;
    const chv_dev_plane &src = a.inputs[0].planes[0];
    for (int c = 0; c < 3; c++) chv_write(dst, x, y, c, 1.0f - chv_read(src, x, y, c));
    chv_write(dst, x, y, 3, a.current.n_planes ? chv_read(a.current.planes[0], x, y, 3) : 1.0f);
}
'''

# the reference's img_nv12_nv12 luma path written against the prelude: Khronos LINEAR sample of plane 0 at uv,
# geometry from the ImageUniforms bytes
SCALE_LUMA = r'''
extern "C" __global__ void scale_luma(chv_custom_args a) {
    const chv_dev_plane &dst = a.target.planes[0];
    CHV_GUARD(dst);
    const ImageUniforms *u = (const ImageUniforms *)a.uniforms;
    float ox = (float)CHV_GID_X / (float)dst.width, oy = (float)CHV_GID_Y / (float)dst.height;
    float4 tx = chv_vecmat4(make_float4(ox * 2.f - 1.f, oy * 2.f - 1.f, 0.f, 1.f), u->transform);
    float4 uv = chv_vecmat4(tx, u->textureTransform);
    chv_write(dst, CHV_GID_X, CHV_GID_Y, 0, chv_sample(a.inputs[0].planes[0], uv.x, uv.y, 0));
}
'''


def test_custom_kernel_build_run_and_library_semantics(ctx):
    assert b"chv_custom_args" in cv.load().chv_custom_prelude()
    w, h = 70, 33                                            # not a multiple of the 16x16 launch blocks
    src = util.alloc_image("bgra", w, h, seed=5)
    dst0 = util.alloc_image("bgra", w, h, seed=6)
    gsrc, gdst = G.to_gpu(ctx, "bgra", w, h, src), G.to_gpu(ctx, "bgra", w, h, dst0)
    k = sv.CustomKernel("invert_bgra")
    with pytest.raises(sv.ComputeError) as e:               # not built yet
        sv.runComputeKernel(ctx, [gsrc], gdst, k)
    assert e.value.case == "computeKernelNotFound"
    sv.buildComputeKernel(ctx, "invert_bgra", INVERT)
    sv.usingContext(ctx, lambda c: sv.runComputeKernel(c, [gsrc], gdst, k, blends=True))
    exp = [np.concatenate([255 - src[0][..., :3], dst0[0][..., 3:]], axis=-1)]
    G.assert_same(G.from_gpu(ctx, gdst, "bgra", w, h), exp, "invert, blends")
    sv.usingContext(ctx, lambda c: sv.runComputeKernel(c, [gsrc], gdst, k, blends=False))
    exp[0][..., 3] = 255
    G.assert_same(G.from_gpu(ctx, gdst, "bgra", w, h), exp, "invert, no current image")
    # a context sharing this one inherits the library as it is now; later builds stay private to their context
    shared = sv.createComputeContext(sharing=ctx)
    sv.usingContext(shared, lambda c: sv.runComputeKernel(c, [gsrc], gdst, k))
    sv.buildComputeKernel(shared, "scale_luma", SCALE_LUMA)
    with pytest.raises(sv.ComputeError) as e:
        sv.runComputeKernel(ctx, [gsrc], gdst, sv.CustomKernel("scale_luma"))
    assert e.value.case == "computeKernelNotFound"
    # compile errors: badInputData carrying the build log; the context stays usable; rebuilding replaces
    with pytest.raises(sv.ComputeError) as e:
        sv.buildComputeKernel(ctx, "broken", 'extern "C" __global__ void broken(chv_custom_args a) { int x = ; }')
    assert e.value.case == "badInputData" and "Unable to create kernel named broken" in str(e.value) and "error" in str(e.value)
    with pytest.raises(sv.ComputeError) as e:               # compiles, but defines no kernel of that name
        sv.buildComputeKernel(ctx, "missing", "__device__ int f() { return 1; }")
    assert e.value.case == "badInputData"
    sv.buildComputeKernel(ctx, "invert_bgra", INVERT.replace("1.0f - chv_read(src, x, y, c)", "chv_read(src, x, y, c)"))
    sv.usingContext(ctx, lambda c: sv.runComputeKernel(c, [gsrc], gdst, k))
    exp[0][..., :3] = src[0][..., :3]
    G.assert_same(G.from_gpu(ctx, gdst, "bgra", w, h), exp, "rebuilt kernel replaces the old one")
    sv.destroyComputeContext(shared)


def test_custom_kernel_with_uniforms_matches_reference_sampler(ctx):
    """A custom kernel written against the prelude's sampler reproduces the luma plane of the reference kernel
    img_nv12_nv12 (Khronos LINEAR, unit-scale arithmetic) bit for bit."""
    (W, H), (w, h) = (96, 54), (64, 36)
    src = util.alloc_image("nv12", W, H, seed=9)
    u = util.full_canvas_uniforms((w, h), (W, H))
    exp = util.alloc_image("nv12", w, h)
    assert O.run_kernel("img_clear_nv12", exp) == 0
    assert O.run_kernel("img_nv12_nv12", exp, src, u) == 0
    gsrc = G.to_gpu(ctx, "nv12", W, H, src)
    gdst = G.to_gpu(ctx, "nv12", w, h, util.alloc_image("nv12", w, h))
    sv.buildComputeKernel(ctx, "scale_luma", SCALE_LUMA)
    sv.usingContext(ctx, lambda c: sv.runComputeKernel(c, [gsrc], gdst, sv.CustomKernel("scale_luma"), uniforms=u))
    got = G.from_gpu(ctx, gdst, "nv12", w, h)
    assert np.array_equal(got[0], exp[0])
    # argument checks
    with pytest.raises(sv.ComputeError):
        sv.runComputeKernel(ctx, [gsrc] * 5, gdst, sv.CustomKernel("scale_luma"), uniforms=u)
    with pytest.raises(sv.ComputeError):
        sv.runComputeKernel(ctx, [gsrc], gdst, sv.CustomKernel("scale_luma"), uniforms=bytes(300))
