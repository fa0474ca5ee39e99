"""The caller side of the path: PictureAnimator -> VideoMixer.  CPU part: the geometry the animator
produces; GPU part: an aspect-fit, bordered, half-transparent layer through the mixer equals the oracle
run with the uniforms applyComputeImage derives from the animator's matrices."""
import numpy as np
import pytest

import util
from swiftvideo_amd import animator as an
from swiftvideo_amd import compute as sv


def test_texture_matrix_aspect_fit_and_fill():
    # 4:3 picture in a 16:9 rect: fit shrinks x, fill shrinks y (animator.pic.swift:213-223)
    fit = an.computeTextureMatrix((640, 480), (1280, 720), (0, 0), an.ASPECT_FIT)
    sx = (640 / 480) / (1280 / 720)
    assert np.allclose(np.diag(fit)[:2], (sx, 1.0)) and np.isclose(fit[0, 3], (1 - sx) / 2) and fit[1, 3] == 0
    fill = an.computeTextureMatrix((640, 480), (1280, 720), (0.1, 0), an.ASPECT_FILL)
    sy = (1280 / 720) / (640 / 480)
    assert np.allclose(np.diag(fill)[:2], (1.0, sy)) and np.isclose(fill[1, 3], (1 - sy) / 2) and np.isclose(fill[0, 3], 0.1)
    assert np.array_equal(an.computeTextureMatrix((640, 480), (1280, 720), (0, 0), an.ASPECT_NONE), np.eye(4))


def test_picture_state_and_uniform_rows():
    st = an.ElementState(picPos=(100, 50, 0), size=(640, 360), borderSize=(4, 2, 6, 8), fillColor=(1, 0, 0, 0.5),
                         transparency=0.25, picOrigin=an.ORIGIN_TOP_LEFT)
    cs = an.computePictureState((1920, 1080), st)
    assert cs.opacity == 0.75 and cs.fillColor == (1, 0, 0, 0.5)
    # unit quad corners land on the rect / the border rect in canvas pixels
    assert np.allclose(cs.matrix @ [0, 0, 0, 1], [100, 50, 0, 1]) and np.allclose(cs.matrix @ [1, 1, 0, 1], [740, 410, 0, 1])
    assert np.allclose(cs.borderMatrix @ [0, 0, 0, 1], [96, 48, 0, 1]) and np.allclose(cs.borderMatrix @ [1, 1, 0, 1], [746, 418, 0, 1])
    # centred origin shifts by half the size (animator.pic.swift:252)
    c = an.computePictureState((64, 64), an.ElementState(picPos=(100, 100, 0), size=(50, 20), picOrigin=an.ORIGIN_CENTER))
    assert np.allclose(c.matrix @ [0, 0, 0, 1], [75, 90, 0, 1])
    # the animator's output, turned into kernel rows by applyComputeImage, equals the test-suite's builder
    anim = an.PictureAnimator((1280, 720), st)
    pic = sv.createPictureSample((1920, 1080), sv.PixelFormat.nv12)
    tag, stamped = anim(pic)
    assert tag == "just"
    u = sv.imageUniformsFor(stamped, sv.createPictureSample((1280, 720), sv.PixelFormat.nv12)).blob()
    ref = util.make_uniforms((1280, 720), rect=(100, 50, 640, 360), border=(4, 2, 6, 8), fill=(1, 0, 0, 0.5), opacity=0.75,
                             in_size=(1920, 1080))
    assert np.allclose(u, ref, atol=1e-6)
    # transitions interpolate (animator.pic.swift:195-205, 278-306); hidden elements emit nothing
    nxt = an.ElementState(picPos=(300, 250, 0), size=(320, 180), transparency=0.75, picAspect=an.ASPECT_FIT)
    mid = an.computeElementState(st, nxt, 0.5)
    assert mid.picPos == (200, 150, 0) and mid.size == (480, 270) and mid.transparency == 0.5 and mid.picAspect == an.ASPECT_FIT
    assert an.PictureAnimator((64, 36), an.ElementState(hidden=True))(pic)[0] == "nothing"


@pytest.mark.gpu
def test_animator_to_mixer_aspect_fit_letterbox(ctx):
    import gpuutil as G
    from oracle import oracle as O
    canvas = (320, 180)
    # a 4:3 picture, aspect-fit into a 16:9 rect with a border and a fill colour: bars left and right
    state = an.ElementState(picPos=(40, 20, 0), size=(240, 135), picAspect=an.ASPECT_FIT, fillColor=(0.1, 0.8, 0.3, 1.0),
                            borderSize=(3, 3, 3, 3), transparency=0.2)
    src = util.alloc_image("y420p", 160, 120, seed=77)
    pic = sv.pictureFromArrays(sv.PixelFormat.y420p, (160, 120), src, assetId="cam")
    tag, stamped = an.PictureAnimator(canvas, state, revision="cam@1")(pic)
    assert tag == "just"
    mixer = sv.VideoMixer("ws", 1 / 30, canvas, outputFormat=sv.PixelFormat.y420p, computeContext=ctx)
    tag, gpu = sv.GPUBarrierUpload(ctx)(stamped)
    mixer.push(gpu)
    out = mixer.mix(at=0.0)
    assert out is not None, mixer.result
    exp = util.alloc_image("y420p", *canvas)
    assert O.run_kernel("img_clear_y420p", exp) == 0
    u = sv.imageUniformsFor(stamped, out).blob()
    assert O.run_kernel("img_y420p_y420p", exp, src, u) == 0
    got = G.from_gpu(ctx, out, "y420p", *canvas)
    G.assert_same(got, exp, "animator -> mixer")
    # letterbox bars (inside the rect, outside the picture) carry the fill colour blended at opacity*alpha
    bar = got[0][60, 45]
    pic_px = got[0][60, 160]
    assert bar == exp[0][60, 45] and bar != 0 and pic_px == exp[0][60, 160]
    assert got[0][5, 5] == 0                       # outside the border quad: cleared canvas
