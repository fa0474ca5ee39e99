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


def test_parent_anchors_follow_the_parents_corners():
    """animator.pic.swift:149-193: a child's corners follow the parent's corners named by its anchors
    when the parent is resized after the child was attached.  dx, dy = growth of the parent."""
    base, size, ppos, d = (10.0, 20.0, 0.0), (100.0, 50.0), (5.0, 7.0), (30.0, 16.0)
    rel = (15.0, 27.0)
    TL, TR, BL, BR = an.ANCHOR_TOP_LEFT, an.ANCHOR_TOP_RIGHT, an.ANCHOR_BOTTOM_LEFT, an.ANCHOR_BOTTOM_RIGHT
    cases = {
        (TL,): (rel, size),                                              # pinned to the parent's top-left: unchanged
        (TR,): ((rel[0] + 30, rel[1]), size),                            # rides the right edge
        (BL,): ((rel[0], rel[1] + 16), size),                            # rides the bottom edge
        (BR,): ((rel[0] + 30, rel[1] + 16), size),                       # rides the bottom-right corner
        (TL, TR): (rel, (130.0, 50.0)),                                  # stretches horizontally
        (TL, BL): (rel, (100.0, 66.0)),                                  # stretches vertically
        (TL, BR): (rel, (130.0, 66.0)),                                  # stretches both ways
        (TR, BR): ((rel[0] + 30, rel[1]), (100.0, 66.0)),                # right edge, stretches vertically
        (BL, BR): ((rel[0], rel[1] + 16), (130.0, 50.0)),                # bottom edge, stretches horizontally
        (TL, TR, BL, BR): (rel, (130.0, 66.0)),
    }
    for anchors, (pos, sz) in cases.items():
        got_pos, got_size = an.computePositionSize(base, size, ppos, d, anchors)
        assert np.allclose(got_pos, pos) and np.allclose(got_size, sz), anchors
    # without growth of the parent every anchor set gives the plain relative rectangle
    for anchors in cases:
        got_pos, got_size = an.computePositionSize(base, size, ppos, (0.0, 0.0), anchors)
        assert np.allclose(got_pos, rel) and np.allclose(got_size, size)


def test_child_animator_tracks_parent_state():
    pic = sv.createPictureSample((64, 64), sv.PixelFormat.BGRA)
    parent = an.PictureAnimator((1280, 720), an.ElementState(picPos=(200, 100, 0), size=(400, 300), transparency=0.5))
    child = an.PictureAnimator((1280, 720), parent=parent)
    assert child(pic)[0] == "nothing"                                     # no state yet
    child.setState(an.ElementState(picPos=(10, 20, 0), size=(100, 50), transparency=0.2,
                                   parentAnchor=(an.ANCHOR_BOTTOM_RIGHT,)))
    assert child.anchors == (an.ANCHOR_BOTTOM_RIGHT,)
    inv = np.linalg.inv(an.orthoMatrix((1280, 720)))
    # reference quirk (animator.pic.swift:115-117, 250): the first sample is computed before the attachment
    # state is recorded, so the parent's whole size counts as growth
    tag, first = child(pic)
    assert tag == "just" and child.initialParentState is not None
    assert np.allclose((inv @ first.matrix()) @ [0, 0, 0, 1], [200 + 10 + 400, 100 + 20 + 300, 0, 1])
    assert np.isclose(first.opacity(), 0.8 * 0.5)                         # own opacity x the parent's
    # from the second sample on: relative to the parent's position, growth measured from the attachment
    tag, second = child(pic)
    assert np.allclose((inv @ second.matrix()) @ [0, 0, 0, 1], [210, 120, 0, 1])
    assert np.allclose((inv @ second.matrix()) @ [1, 1, 0, 1], [310, 170, 0, 1])
    parent.setState(an.ElementState(picPos=(220, 90, 0), size=(460, 330), transparency=0.5))
    tag, third = child(pic)
    assert np.allclose((inv @ third.matrix()) @ [0, 0, 0, 1], [220 + 10 + 60, 90 + 20 + 30, 0, 1])   # rides the corner
    assert np.allclose((inv @ third.matrix()) @ [1, 1, 0, 1], [220 + 10 + 60 + 100, 90 + 20 + 30 + 50, 0, 1])
    # a rotated parent: its size is read off the matrix columns (animator.pic.swift:243-245)
    parent.setState(an.ElementState(picPos=(220, 90, 0), size=(460, 330), rotation=0.3))
    assert np.allclose(an._col_scale(parent.computedState(pic).matrix), (460, 330))
    # an immediate setState re-attaches (initialParentState reset, :56-66)
    child.setState(an.ElementState(picPos=(0, 0, 0), size=(10, 10)))
    assert child.initialParentState is None and child.anchors == (an.ANCHOR_TOP_LEFT,)


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
