"""Host-side mirror of the caller that produces the matrices the kernels consume:
PictureAnimator / computePictureState / computePositionSize / computeTextureMatrix
(animator.pic.swift:28-128,149-193,207-272,326-332), including elements attached to a parent
element through parent anchors.

The 4x4 algebra lives in the un-vendored, un-pinned VectorMath package (Package.swift:61); the
conventions used here are the ones SURVEY section 8c derives and DESIGN.md section 3 lists as unpinned:
column vectors, M = projection * T(pos) * R(z, rotation) * S(size).  No pixel work happens here.
"""
from dataclasses import dataclass, field, replace

import numpy as np

ASPECT_NONE, ASPECT_FIT, ASPECT_FILL = "aspectNone", "aspectFit", "aspectFill"
ORIGIN_TOP_LEFT, ORIGIN_CENTER = "originTopLeft", "originCenter"
ANCHOR_TOP_LEFT, ANCHOR_TOP_RIGHT = "anchorTopLeft", "anchorTopRight"           # Proto/Composition.proto PictureAnchor
ANCHOR_BOTTOM_LEFT, ANCHOR_BOTTOM_RIGHT = "anchorBottomLeft", "anchorBottomRight"


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
    parentAnchor: tuple = ()           # empty -> [anchorTopLeft], animator.pic.swift:64

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


def _col_scale(m):
    """Lengths of the first two columns' xy parts = the element's size under rotation
    (sqrt(m11^2 + m12^2), sqrt(m21^2 + m22^2), animator.pic.swift:243-249; VectorMath m11..m14 = column 1)."""
    return (float(np.hypot(m[0, 0], m[1, 0])), float(np.hypot(m[0, 1], m[1, 1])))


def computePositionSize(basePos, baseSize, parentPos, parentSizeDelta, anchors):
    """animator.pic.swift:149-193: position and size of an element whose corners follow its parent's
    corners.  basePos/baseSize: the element's own state; parentPos: the parent's translation;
    parentSizeDelta: parent size now minus parent size when the element was attached."""
    rel = (basePos[0] + parentPos[0], basePos[1] + parentPos[1])
    dx, dy = parentSizeDelta[0], parentSizeDelta[1]
    v = [[rel[0], rel[1]], [rel[0] + baseSize[0], rel[1]], [rel[0], rel[1] + baseSize[1]]]
    a = set(anchors)
    if ANCHOR_BOTTOM_RIGHT in a:
        v = [[x + dx, y + dy] for x, y in v]
        if ANCHOR_BOTTOM_LEFT in a:
            v[0][0] = rel[0]
            v[2][0] = rel[0]
        if ANCHOR_TOP_RIGHT in a:
            v[0][1] = rel[1]
            v[1][1] = rel[1]
        if ANCHOR_TOP_LEFT in a:
            v[0] = [rel[0], rel[1]]
            v[1] = [rel[0] + baseSize[0] + dx, rel[1]]
            v[2] = [rel[0], rel[1] + baseSize[1] + dy]
    elif ANCHOR_TOP_RIGHT in a:
        v[1][0] += dx
        if ANCHOR_TOP_LEFT not in a and ANCHOR_BOTTOM_LEFT not in a:
            v[0][0] += dx
            v[2][0] += dx
        elif ANCHOR_BOTTOM_LEFT in a:
            v[2][1] += dy
    elif ANCHOR_BOTTOM_LEFT in a:
        v[2][1] += dy
        if ANCHOR_TOP_LEFT not in a:
            v[1][1] += dy
            v[0][1] += dy
    return (v[0][0], v[0][1]), (v[1][0] - v[0][0], v[2][1] - v[0][1])


def computePictureState(sampleSize, current, next=None, pct=None, parent=None, anchors=(ANCHOR_TOP_LEFT,),
                        initialParentState=None):
    """animator.pic.swift:229-272.  parent: the parent's (un-projected) matrix or None."""
    state = computeElementState(current, next, pct) if (next is not None and pct is not None) else current
    parentPos, parentSize = ((float(parent[0, 3]), float(parent[1, 3])), _col_scale(parent)) if parent is not None else ((0.0, 0.0), (0.0, 0.0))
    initialParentSize = _col_scale(initialParentState.matrix) if initialParentState is not None else (0.0, 0.0)
    parentSizeDelta = (parentSize[0] - initialParentSize[0], parentSize[1] - initialParentSize[1])
    add = (0.0, 0.0) if state.picOrigin == ORIGIN_TOP_LEFT else (-state.size[0] / 2, -state.size[1] / 2)
    relPos, size = computePositionSize(state.picPos, state.size, parentPos, parentSizeDelta, anchors)
    pos = (relPos[0] + add[0], relPos[1] + add[1])
    bl, bt, br, bb = state.borderSize
    borderPos = (pos[0] - bl, pos[1] - bt)
    borderSize = (bl + size[0] + br, bt + size[1] + bb)
    return ComputedPictureState(
        matrix=_translation(*pos) @ _rotation_z(state.rotation) @ _scale(*size),
        textureMatrix=computeTextureMatrix(sampleSize, size, state.textureOffset, state.picAspect),
        borderMatrix=_translation(*borderPos) @ _rotation_z(state.rotation) @ _scale(*borderSize),
        fillColor=state.getFillColor(), opacity=1.0 - state.transparency)


class AnimatorError(Exception):
    """animator.pic.swift:20-22 (noCurrentState)"""


class PictureAnimator:
    """Tx<PictureSample, PictureSample>, animator.pic.swift:28-128: stamps each sample with
    projection * matrix, textureMatrix, projection * borderMatrix, fill colour, opacity, revision.
    With a parent animator the element's corners follow the parent's (parentAnchors) and its opacity
    is multiplied by the parent's."""

    def __init__(self, canvasSize, state=None, revision=None, parentOpacity=1.0, parent=None,
                 parentAnchors=(ANCHOR_TOP_LEFT,)):
        self.canvasSize = canvasSize
        self.currentState = state
        self.nextState = None
        self.pct = None
        self.revision = revision
        self.parentOpacity = parentOpacity
        self.parent = parent
        self.anchors = tuple(parentAnchors)
        self.initialParentState = None

    def setParent(self, parent):
        self.parent = parent

    def setState(self, nxt, pct=None):
        """pct None: switch immediately (duration <= 0 in the reference: anchors follow the new state,
        the parent attachment is re-initialised, animator.pic.swift:56-66); otherwise a transition that
        is `pct` of the way through."""
        if pct is None or self.currentState is None:
            self.currentState, self.nextState, self.pct = nxt, None, None
            self.initialParentState = None
            self.anchors = tuple(nxt.parentAnchor) if len(nxt.parentAnchor) > 0 else (ANCHOR_TOP_LEFT,)
        else:
            self.nextState, self.pct = nxt, pct

    def finishTransition(self):
        """the scheduled completion of a timed transition, animator.pic.swift:70-78"""
        if self.nextState is None:
            return
        self.anchors = tuple(self.nextState.parentAnchor)
        self.currentState, self.nextState, self.pct = self.nextState, None, None
        self.initialParentState = None

    def computedState(self, sample, parentState=None):
        """animator.pic.swift:84-105"""
        if self.currentState is None:
            raise AnimatorError("noCurrentState")
        return computePictureState(sample.size(), self.currentState, self.nextState, self.pct,
                                   parent=parentState.matrix if parentState is not None else None,
                                   anchors=self.anchors, initialParentState=self.initialParentState)

    def __call__(self, sample):
        if self.currentState is None or self.currentState.hidden:
            return ("nothing", sample.info())
        try:
            # the parent's state is computed without ITS parent (animator.pic.swift:112), and the state a
            # child was attached under is recorded only after the first sample went through (:115-117)
            parentState = self.parent.computedState(sample) if self.parent is not None else None
            cs = self.computedState(sample, parentState)
        except AnimatorError:
            return ("nothing", sample.info())
        opacity = parentState.opacity if parentState is not None else 1.0
        if parentState is not None and self.initialParentState is None:
            self.initialParentState = parentState
        proj = orthoMatrix(self.canvasSize)
        kw = dict(matrix=proj @ cs.matrix, textureMatrix=cs.textureMatrix, borderMatrix=proj @ cs.borderMatrix,
                  fillColor=cs.fillColor, opacity=cs.opacity * opacity * self.parentOpacity)
        if self.revision is not None:
            kw["revision"] = self.revision
        return ("just", sample.derive(**kw))
