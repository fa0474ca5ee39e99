"""Shared test helpers: deterministic inputs, ImageUniforms builders, plane allocation."""
import numpy as np

MASK = (1 << 64) - 1


def splitmix_bytes(seed, n):
    """Low byte of successive splitmix64 outputs, vectorised."""
    idx = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = np.uint64(seed & MASK) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z & np.uint64(0xFF)).astype(np.uint8)


# ---- planes -----------------------------------------------------------------
def plane_shapes(fmt, w, h):
    """[(rows, cols, comps)] per plane, layouts of sample.pict.linux.swift:275-294."""
    fmt = fmt.lower()
    if fmt == "nv12":
        return [(h, w, 1), (h // 2, w // 2, 2)]
    if fmt == "y420p":
        return [(h, w, 1), (h // 2, w // 2, 1), (h // 2, w // 2, 1)]
    if fmt in ("bgra", "rgba"):
        return [(h, w, 4)]
    raise ValueError(fmt)


def alloc_image(fmt, w, h, seed=None, pad=0):
    """List of uint8 plane arrays (h,w) or (h,w,c); optional random fill; `pad` extra pitch bytes."""
    planes = []
    for k, (r, c, comps) in enumerate(plane_shapes(fmt, w, h)):
        r = max(r, 1)
        c = max(c, 1)
        pitch = c * comps + pad
        buf = np.zeros((r, pitch), dtype=np.uint8)
        if seed is not None:
            buf[:] = splitmix_bytes(seed * 16 + k, r * pitch).reshape(r, pitch)
        view = buf[:, : c * comps]
        view = view if comps == 1 else view.reshape(r, c, comps)
        planes.append(view)
    return planes


def copy_image(planes):
    out = []
    for p in planes:
        base = p.base if p.base is not None else p
        # keep the pitch of the original
        pitch = p.strides[0]
        buf = np.zeros((p.shape[0], pitch), dtype=np.uint8)
        comps = 1 if p.ndim == 2 else p.shape[2]
        view = buf[:, : p.shape[1] * comps]
        view = view if comps == 1 else view.reshape(p.shape[0], p.shape[1], comps)
        view[...] = p
        out.append(view)
    return out


# ---- uniforms ---------------------------------------------------------------
def _mat_translate(x, y):
    m = np.eye(4)
    m[0, 3], m[1, 3] = x, y
    return m


def _mat_scale(x, y):
    return np.diag([x, y, 1.0, 1.0])


def _mat_rot(theta):
    c, s = np.cos(theta), np.sin(theta)
    m = np.eye(4)
    m[0, 0], m[0, 1], m[1, 0], m[1, 1] = c, -s, s, c
    return m


def ortho(cw, ch):
    """Matrix4(ortho) of animator.pic.swift:326-332 as a column-vector matrix: canvas px -> NDC."""
    m = np.eye(4)
    m[0, 0], m[1, 1] = 2.0 / cw, 2.0 / ch
    m[0, 3], m[1, 3] = -1.0, -1.0
    m[2, 3] = 1.0
    return m


def make_uniforms(canvas, rect=None, rotation=0.0, border=(0, 0, 0, 0), tex=None,
                  fill=(0, 0, 0, 0), opacity=1.0, in_size=(0, 0)):
    """59-float ImageUniforms blob (compute.swift:76-86).

    Kernel row i of each matrix is row i of M^-1 (compute.swift:151-155 uploads
    M.inverse.transpose), M = ortho * T(pos) * R(rot) * S(size) as in
    animator.pic.swift:118-123,264-267.  rect = (x, y, w, h) in canvas pixels.
    tex = (ox, oy, sx, sy): texture matrix T(ox,oy)*S(sx,sy) (animator.pic.swift:217-223).
    """
    cw, ch = canvas
    if rect is None:
        rect = (0, 0, cw, ch)
    x, y, w, h = rect
    M = ortho(cw, ch) @ _mat_translate(x, y) @ _mat_rot(rotation) @ _mat_scale(w, h)
    bl, bt, br, bb = border
    B = ortho(cw, ch) @ _mat_translate(x - bl, y - bt) @ _mat_rot(rotation) @ _mat_scale(bl + w + br, bt + h + bb)
    T = np.eye(4)
    if tex is not None:
        T = _mat_translate(tex[0], tex[1]) @ _mat_scale(tex[2], tex[3])
    u = np.zeros(59, dtype=np.float32)
    u[0:16] = np.linalg.inv(M).astype(np.float32).reshape(-1)
    u[16:32] = np.linalg.inv(T).astype(np.float32).reshape(-1)
    u[32:48] = np.linalg.inv(B).astype(np.float32).reshape(-1)
    u[48:52] = fill
    u[52:54] = in_size
    u[54:56] = canvas
    u[56] = opacity
    return u


def full_canvas_uniforms(canvas, in_size, opacity=1.0, fill=(0, 0, 0, 0)):
    """The literal full-canvas rows of SURVEY section 8c."""
    u = np.zeros(59, dtype=np.float32)
    rows = [(.5, 0, 0, .5), (0, .5, 0, .5), (0, 0, 1, -1), (0, 0, 0, 1)]
    u[0:16] = np.array(rows, dtype=np.float32).reshape(-1)
    u[16:32] = np.eye(4, dtype=np.float32).reshape(-1)
    u[32:48] = u[0:16]
    u[48:52] = fill
    u[52:54] = in_size
    u[54:56] = canvas
    u[56] = opacity
    return u
