"""Host-side logic that runs without a GPU: plane layouts, uniforms construction, mixer bookkeeping."""
import numpy as np
import pytest

import util
from swiftvideo_amd import compute as sv


def test_planes_for_format_follow_reference_layouts():
    # sample.pict.linux.swift:275-294
    p = sv.planesForFormat(sv.PixelFormat.nv12, (1920, 1080))
    assert [(x.size, x.stride, len(x.components)) for x in p] == [((1920, 1080), 1920, 1), ((960, 540), 1920, 2)]
    p = sv.planesForFormat(sv.PixelFormat.y420p, (1280, 720))
    assert [(x.size, x.stride) for x in p] == [((1280, 720), 1280), ((640, 360), 640), ((640, 360), 640)]
    p = sv.planesForFormat(sv.PixelFormat.BGRA, (33, 17))
    assert [(x.size, x.stride, len(x.components)) for x in p] == [((33, 17), 132, 4)]
    with pytest.raises(sv.ComputeError) as e:
        sv.planesForFormat(sv.PixelFormat.yuvs + 100 if False else sv.PixelFormat.nv21, (4, 4))
    assert e.value.case == "badInputData"


def test_create_picture_sample():
    s = sv.createPictureSample((64, 36), sv.PixelFormat.nv12, assetId="a", workspaceId="w")
    assert s.bufferType() == "cpu" and s.pixelFormat() == sv.PixelFormat.nv12 and s.size() == (64, 36)
    assert [b.shape for b in s.imageBuffer().buffers] == [(36, 64), (18, 64)]
    assert all(not b.any() for b in s.imageBuffer().buffers)
    with pytest.raises(sv.ComputeError) as e:
        sv.createPictureSample((0, 10), sv.PixelFormat.nv12)
    assert e.value.case == "invalidOperation"            # sample.pict.linux.swift:259-261
    with pytest.raises(sv.ComputeError):
        sv.ImageBuffer(sv.PixelFormat.nv12, "cpu", (4, 4))   # neither textures nor buffers


def test_image_uniforms_blob_layout():
    u = sv.ImageUniforms(transform=np.arange(16).reshape(4, 4), textureTransform=np.eye(4), borderMatrix=np.eye(4) * 2,
                         fillColor=(0.1, 0.2, 0.3, 0.4), inputSize=(640, 360), outputSize=(1280, 720), opacity=0.5,
                         imageTime=1.25, targetTime=2.5).blob()
    assert u.dtype == np.float32 and u.nbytes == 236
    assert u[0:16].tolist() == list(range(16)) and u[20] == 0 and u[21] == 1 and u[32] == 2
    assert np.allclose(u[48:52], (0.1, 0.2, 0.3, 0.4)) and u[52:56].tolist() == [640, 360, 1280, 720]
    assert u[56] == 0.5 and u[57] == 1.25 and u[58] == 2.5


def test_apply_compute_image_uniforms_are_inverse_rows():
    """compute.swift:149-155 uploads M.inverse.transpose; kernel row i must give (M^-1 v)_i."""
    canvas, rect = (1280, 720), (100, 50, 640, 360)
    M = util.ortho(*canvas) @ util._mat_translate(rect[0], rect[1]) @ util._mat_scale(rect[2], rect[3])
    img = sv.PictureSample(sv.ImageBuffer(sv.PixelFormat.BGRA, "cpu", (64, 36), buffers=[np.zeros((36, 256), np.uint8)],
                                          planes=sv.planesForFormat(sv.PixelFormat.BGRA, (64, 36))), matrix=M, opacity=0.25)
    tgt = sv.createPictureSample(canvas, sv.PixelFormat.nv12)
    u = sv.imageUniformsFor(img, tgt).blob()
    rows = u[0:16].reshape(4, 4).astype(np.float64)
    # the rect's top-left canvas pixel maps to tx = (0, 0), its bottom-right to (1, 1)
    for (px, py), exp in (((100, 50), (0, 0)), ((740, 410), (1, 1)), ((420, 230), (0.5, 0.5))):
        ndc = np.array([px / canvas[0] * 2 - 1, py / canvas[1] * 2 - 1, 0, 1])
        assert np.allclose((rows @ ndc)[:2], exp, atol=1e-5)
    assert np.allclose(u[0:16], util.make_uniforms(canvas, rect=rect)[0:16], atol=1e-6)
    assert u[56] == 0.25 and u[52:56].tolist() == [64, 36, 1280, 720]
    # and the literal full-canvas rows of SURVEY section 8c
    full = util.make_uniforms((64, 36))
    assert np.allclose(full[0:16], util.full_canvas_uniforms((64, 36), (1, 1))[0:16], atol=1e-7)


def test_find_kernel_names():
    class M(sv.VideoMixer):
        def __init__(self, family):
            self.bgraKernelFamily = family
    nv = sv.createPictureSample((8, 8), sv.PixelFormat.nv12)
    bg = sv.createPictureSample((8, 8), sv.PixelFormat.BGRA)
    yp = sv.createPictureSample((8, 8), sv.PixelFormat.y420p)
    m = M("tx")
    assert str(m.findKernel(None, nv)) == "img_clear_nv12" and str(m.findKernel(None, bg)) == "img_clear_bgra"
    assert str(m.findKernel(bg, nv)) == "img_bgra_nv12" and str(m.findKernel(yp, yp)) == "img_y420p_y420p"
    assert str(m.findKernel(nv, bg)) == "img_nv12_bgra" and str(m.findKernel(bg, bg)) == "img_bgra_bgra_tx"
    assert str(M("reference").findKernel(bg, bg)) == "img_bgra_bgra"       # the unchanged Swift VideoMixer: Metal semantics
    assert sv.VideoMixer.__init__.__defaults__[sv.VideoMixer.__init__.__code__.co_varnames.index("bgraKernelFamily") - 4] == "reference"
    with pytest.raises(sv.ComputeError):
        m.findKernel(nv, yp)                                  # img_nv12_y420p: no such kernel in the reference either


def test_splitmix_is_the_documented_generator():
    # z=(x+=0x9E37...); z=(z^(z>>30))*0xBF58...; z=(z^(z>>27))*0x94D0...; z^=z>>31 ; low byte
    def ref(seed, n):
        x, out = seed, []
        for _ in range(n):
            x = (x + 0x9E3779B97F4A7C15) & util.MASK
            z = x
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & util.MASK
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & util.MASK
            z ^= z >> 31
            out.append(z & 0xFF)
        return out
    assert util.splitmix_bytes(0x5EED0000, 64).tolist() == ref(0x5EED0000, 64)


def test_uniform_blobs_of_picture_kernels_stay_float32_whatever_array_comes_in():
    """ImageUniforms are 59 floats: a list or an integer array of matrix rows converts to float32 as it always did; raw bytes pass for the value
    types that are not ImageUniforms — the buffer kernels' BufferUniforms / MotionEstimationUniforms and .custom kernels (`raw`), or explicit bytes"""
    import numpy as np
    from swiftvideo_amd import compute as sv
    ints = np.arange(59, dtype=np.int64)
    b = sv._uniform_blob(ints)
    assert b.dtype == np.float32 and b.nbytes == 236 and b[7] == 7.0
    assert sv._uniform_blob([0.5] * 59).dtype == np.float32
    assert sv._uniform_blob(np.arange(59, dtype=np.int32)).dtype == np.float32            # int32 too, unless the kernel takes raw words
    raw = sv._uniform_blob(np.arange(25, dtype=np.int32), raw=True)
    assert raw.dtype == np.int32 and raw.nbytes == 100
    assert sv._uniform_blob(np.zeros(4, dtype=np.float64), raw=True).dtype == np.float32    # floats are never "raw"
    assert sv._uniform_blob(b"\x01\x02\x03").tolist() == [1, 2, 3]
    assert sv._uniform_blob(None) is None
