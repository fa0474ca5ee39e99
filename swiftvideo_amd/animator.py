"""Host-side mirror of the caller that produces the matrices the kernels consume:
PictureAnimator / computePictureState / computeTextureMatrix (animator.pic.swift:107-128,207-272,
326-332) for elements without a parent (parent anchors only matter relative to a parent's size
change, animator.pic.swift:149-193).

The 4x4 algebra lives in the un-vendored, un-pinned VectorMath package (Package.swift:61); the
conventions used here are the ones SURVEY section 8c derives and DESIGN.md section 3 lists as unpinned:
column vectors, M = projection * T(pos) * R(z, rotation) * S(size).  No pixel work happens here.
"""
from dataclasses import dataclass, field, replace

import numpy as np

ASPECT_NONE, ASPECT_FIT, ASPECT_FILL = "aspectNone", "aspectFit", "aspectFill"
ORIGIN_TOP_LEFT, ORIGIN_CENTER = "originTopLeft", "originCenter"


@dataclass
class ElementState:
    """Proto/Composition.proto:56-71 (the picture fields)."""
    picPos: tuple = (0.0, 0.0, 0.0)
    size: tuple = (0.0, 0.0)
    textureOffset: tuple = (0.0, 0.0)
    rotation: float = 0.0
    transparency: float = 0.0
    picAspect: str = ASPECT_NONE
    picOrigin: str = ORIGIN_TOP_LEFT
    fillColor: tuple = None            # r g b a; None = hasFillColor false -> (0,0,0,0), animator.pic.swift:334-342
    borderSize: tuple = (0.0, 0.0, 0.0, 0.0)   # l t r b
    hidden: bool = False

    def getFillColor(self):
        return tuple(self.fillColor) if self.fillColor is not None else (0.0, 0.0, 0.0, 0.0)


@dataclass
class ComputedPictureState:
    matrix: np.ndarray
    textureMatrix: np.ndarray
    borderMatrix: np.ndarray
    fillColor: tuple
    opacity: float


def _translation(x, y, z=0.0):
    m = np.eye(4)
    m[0, 3], m[1, 3], m[2, 3] = x, y, z
    return m


def _scale(x, y, z=1.0):
    return np.diag([x, y, z, 1.0])


def _rotation_z(t):
    c, s = np.cos(t), np.sin(t)
    m = np.eye(4)
    m[0, 0], m[0, 1], m[1, 0], m[1, 1] = c, -s, s, c
    return m


def orthoMatrix(canvasSize):
    """Matrix4(ortho), animator.pic.swift:326-332: canvas pixels -> NDC."""
    m = np.eye(4)
    m[0, 0], m[1, 1] = 2.0 / canvasSize[0], 2.0 / canvasSize[1]
    m[0, 3], m[1, 3], m[2, 3] = -1.0, -1.0, 1.0
    return m


def interpolate(a, b, pct):
    """animator.pic.swift:278-306"""
    if isinstance(a, (tuple, list)):
        return tuple(x + (y - x) * pct for x, y in zip(a, b))
    return a + (b - a) * pct


def computeElementState(current, nxt, pct):
    """animator.pic.swift:195-205: position/size/offset/rotation/transparency/fill/border interpolate,
    aspect and origin switch to the next state's."""
    return replace(current,
                   picPos=interpolate(current.picPos, nxt.picPos, pct), size=interpolate(current.size, nxt.size, pct),
                   textureOffset=interpolate(current.textureOffset, nxt.textureOffset, pct),
                   rotation=interpolate(current.rotation, nxt.rotation, pct),
                   transparency=interpolate(current.transparency, nxt.transparency, pct),
                   picAspect=nxt.picAspect, picOrigin=nxt.picOrigin,
                   fillColor=interpolate(current.getFillColor(), nxt.getFillColor(), pct),
                   borderSize=interpolate(current.borderSize, nxt.borderSize, pct))


def computeTextureMatrix(sampleSize, geometrySize, textureOffset, aspect):
    """animator.pic.swift:207-227"""
    origAspect = sampleSize[0] / sampleSize[1]
    geomAspect = geometrySize[0] / geometrySize[1]
    if aspect == ASPECT_FIT:
        scalex = 1.0 if origAspect > geomAspect else origAspect / geomAspect
        scaley = 1.0 if origAspect <= geomAspect else geomAspect / origAspect
    elif aspect == ASPECT_FILL:
        scalex = 1.0 if origAspect <= geomAspect else origAspect / geomAspect
        scaley = 1.0 if origAspect > geomAspect else geomAspect / origAspect
    else:
        return np.eye(4)
    return _translation(textureOffset[0] + (1.0 - scalex) / 2, textureOffset[1] + (1.0 - scaley) / 2) @ _scale(scalex, scaley)


def computePictureState(sampleSize, current, next=None, pct=None):
    """animator.pic.swift:229-272 with parent == nil."""
    state = computeElementState(current, next, pct) if (next is not None and pct is not None) else current
    add = (0.0, 0.0) if state.picOrigin == ORIGIN_TOP_LEFT else (-state.size[0] / 2, -state.size[1] / 2)
    pos = (state.picPos[0] + add[0], state.picPos[1] + add[1])
    size = (state.size[0], state.size[1])
    bl, bt, br, bb = state.borderSize
    borderPos = (pos[0] - bl, pos[1] - bt)
    borderSize = (bl + size[0] + br, bt + size[1] + bb)
    return ComputedPictureState(
        matrix=_translation(*pos) @ _rotation_z(state.rotation) @ _scale(*size),
        textureMatrix=computeTextureMatrix(sampleSize, size, state.textureOffset, state.picAspect),
        borderMatrix=_translation(*borderPos) @ _rotation_z(state.rotation) @ _scale(*borderSize),
        fillColor=state.getFillColor(), opacity=1.0 - state.transparency)


class PictureAnimator:
    """Tx<PictureSample, PictureSample>, animator.pic.swift:24-128: stamps each sample with
    projection * matrix, textureMatrix, projection * borderMatrix, fill colour, opacity, revision."""

    def __init__(self, canvasSize, state, revision=None, parentOpacity=1.0):
        self.canvasSize = canvasSize
        self.currentState = state
        self.nextState = None
        self.pct = None
        self.revision = revision
        self.parentOpacity = parentOpacity

    def setState(self, nxt, pct=None):
        """pct None: switch immediately; otherwise a transition that is `pct` of the way through."""
        if pct is None:
            self.currentState, self.nextState, self.pct = nxt, None, None
        else:
            self.nextState, self.pct = nxt, pct

    def __call__(self, sample):
        if self.currentState is None or self.currentState.hidden:
            return ("nothing", sample.info())
        cs = computePictureState(sample.size(), self.currentState, self.nextState, self.pct)
        proj = orthoMatrix(self.canvasSize)
        kw = dict(matrix=proj @ cs.matrix, textureMatrix=cs.textureMatrix, borderMatrix=proj @ cs.borderMatrix,
                  fillColor=cs.fillColor, opacity=cs.opacity * self.parentOpacity)
        if self.revision is not None:
            kw["revision"] = self.revision
        return ("just", sample.derive(**kw))
