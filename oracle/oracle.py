"""ctypes front-end of the CPU checker (oracle/liboracle.so, oracle/_ref/libclref.so).

TEST INFRASTRUCTURE, NOT PRODUCT.  Importable only from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.  PARITY UNPINNED — see
oracle/ref_kernels.h.
"""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent

# kernel ids (mirror of ref_kernels.h)
KERNEL_IDS = {
    "img_nv12_nv12": 0, "img_bgra_nv12": 1, "img_rgba_nv12": 2, "img_bgra_bgra": 3,
    "img_y420p_y420p": 4, "img_y420p_nv12": 5, "img_clear_nv12": 6, "img_clear_yuvs": 7,
    "img_clear_bgra": 8, "img_clear_y420p": 9, "img_clear_rgba": 10, "img_rgba_y420p": 11,
    "img_bgra_y420p": 12, "snd_s16i_s16i": 13, "me_fullsearch": 14,
    "img_nv12_bgra": 32, "img_y420p_bgra": 33, "img_bgra_bgra_tx": 34, "img_rgba_bgra_tx": 35,
    "img_bgra_nv12_int": 36, "img_rgba_nv12_int": 37, "img_bgra_y420p_int": 38, "img_rgba_y420p_int": 39,
}
# unit-scale envelope evaluators of the BGRA-target family (tests only; ref_kernels.c::px_to_bgra_unit)
ENVELOPE_IDS = {"img_nv12_bgra": 64, "img_y420p_bgra": 65, "img_bgra_bgra_tx": 66, "img_rgba_bgra_tx": 67}
CSC = {"bt601": 0, "bt709": 1, "bt601_full": 2, "bt709_full": 3}


class Plane(C.Structure):
    _fields_ = [("data", C.c_void_p), ("w", C.c_int32), ("h", C.c_int32),
                ("pitch", C.c_int32), ("comps", C.c_int32)]


def build(force=False):
    """Compile the restatement (and the clref cross-check when /root/reference exists)."""
    did = {"liboracle": "reused", "clref": "absent"}
    so = HERE / "liboracle.so"
    if force or not so.exists() or so.stat().st_mtime < (HERE / "ref_kernels.c").stat().st_mtime:
        subprocess.check_call(["make", "-C", str(HERE), "liboracle.so"], stdout=subprocess.DEVNULL)
        did["liboracle"] = "compiled"
    ref = HERE / "_ref" / "libclref.so"
    if os.path.exists("/root/reference/Sources/SwiftVideo/kernels.cl.swift"):
        if force or not ref.exists():
            subprocess.check_call(["make", "-C", str(HERE), "clref"], stdout=subprocess.DEVNULL)
            did["clref"] = "compiled"
        else:
            did["clref"] = "reused"
    elif ref.exists():
        did["clref"] = "prebuilt (no reference tree here)"
    return did


_lib = None
_clref = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(HERE / "liboracle.so"))
        _lib.orc_run_kernel.restype = C.c_int
        _lib.orc_run_kernel.argtypes = [C.c_int, C.POINTER(Plane), C.c_int, C.POINTER(Plane), C.c_int,
                                        C.c_void_p, C.c_int, C.c_int]
        _lib.orc_lanczos_bgra.restype = C.c_int
        _lib.orc_lanczos_bgra.argtypes = [C.POINTER(Plane), C.POINTER(Plane), C.c_int]
        _lib.orc_lanczos_table.restype = C.c_int
        _lib.orc_lanczos_table.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_int]
        _lib.orc_store_unorm8.restype = C.c_uint8
        _lib.orc_store_unorm8.argtypes = [C.c_float]
        _lib.orc_load_unorm8.restype = C.c_float
        _lib.orc_load_unorm8.argtypes = [C.c_uint8]
        _lib.orc_yuv2rgb_int.restype = None
        _lib.orc_yuv2rgb_int.argtypes = [C.c_int, C.c_uint8, C.c_uint8, C.c_uint8, C.c_void_p]
        _lib.orc_rgb2yuv_int.restype = None
        _lib.orc_rgb2yuv_int.argtypes = [C.c_int, C.c_uint8, C.c_uint8, C.c_uint8, C.c_void_p]
        _lib.orc_snd_s16i_s16i.restype = C.c_int
        _lib.orc_snd_s16i_s16i.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_void_p]
        _lib.orc_me_fullsearch.restype = C.c_int
        _lib.orc_me_fullsearch.argtypes = [C.POINTER(Plane), C.POINTER(Plane), C.POINTER(Plane), C.c_void_p]
        _lib.orc_me_cost_table.restype = None
        _lib.orc_me_cost_table.argtypes = [C.c_void_p, C.c_int]
    return _lib


def clref():
    """The cross-check library, or None when it has not been built (GPU box without prebuilt file)."""
    global _clref
    if _clref is None:
        p = HERE / "_ref" / "libclref.so"
        if not p.exists():
            try:
                build()
            except Exception:
                return None
        if not p.exists():
            return None
        _clref = C.CDLL(str(p))
        _clref.clref_run.restype = C.c_int
        _clref.clref_run.argtypes = [C.c_char_p, C.POINTER(Plane), C.c_int, C.POINTER(Plane), C.c_int, C.c_void_p]
        if hasattr(_clref, "clref_run_snd"):
            _clref.clref_run_snd.restype = C.c_int
            _clref.clref_run_snd.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_void_p]
    return _clref


def _planes(arrs):
    """arrs: list of uint8 ndarrays shaped (h, w) / (h, w, comps), C-contiguous rows (pitch = strides[0])."""
    out = (Plane * max(1, len(arrs)))()
    for i, a in enumerate(arrs):
        assert a.dtype == np.uint8 and a.strides[-1] == 1
        comps = 1 if a.ndim == 2 else a.shape[2]
        if a.ndim == 3:
            assert a.strides[1] == comps
        out[i] = Plane(a.ctypes.data, a.shape[1], a.shape[0], a.strides[0], comps)
    return out


def run_kernel(name, target, inputs=(), uniforms=None, csc=0, threads=1):
    """Run one kernel in place on `target` (list of plane arrays)."""
    kid = KERNEL_IDS[name] if isinstance(name, str) else int(name)
    t = _planes(list(target))
    i = _planes(list(inputs))
    up = None
    if uniforms is not None:
        u = np.ascontiguousarray(uniforms, dtype=np.float32)
        assert u.size == 59
        up = u.ctypes.data
    rc = lib().orc_run_kernel(kid, t, len(target), i, len(inputs), up, int(csc), int(threads))
    return rc


def run_clref(name, target, inputs=(), uniforms=None):
    l = clref()
    if l is None:
        raise RuntimeError("libclref.so not built")
    t = _planes(list(target))
    i = _planes(list(inputs))
    up = None
    if uniforms is not None:
        u = np.ascontiguousarray(uniforms, dtype=np.float32)
        up = u.ctypes.data
    return l.clref_run(name.encode(), t, len(target), i, len(inputs), up)


def lanczos_bgra(dst, src, threads=1):
    d = _planes([dst])
    s = _planes([src])
    return lib().orc_lanczos_bgra(d, s, int(threads))


def lanczos_table(in_size, out_size, max_taps=256):
    taps = C.c_int(0)
    first = np.zeros(out_size, dtype=np.int32)
    w = np.zeros(out_size * max_taps, dtype=np.float32)
    rc = lib().orc_lanczos_table(in_size, out_size, C.byref(taps), first.ctypes.data, w.ctypes.data, max_taps)
    assert rc == 0
    return taps.value, first, w[: out_size * taps.value].reshape(out_size, taps.value).copy()


def yuv2rgb_int(csc, y, u, v):
    out = (C.c_uint8 * 3)()
    lib().orc_yuv2rgb_int(csc, y, u, v, out)
    return tuple(out)


def rgb2yuv_int(csc, r, g, b):
    out = (C.c_uint8 * 3)()
    lib().orc_rgb2yuv_int(csc, r, g, b, out)
    return tuple(out)


def snd_uniforms(gains, fades, offsets=None):
    """BufferUniforms, kernels.cl.swift:536-541: int inputCount; int inputOffsets[8]; float inputGains[8]; float inputFade[8] (100 bytes)"""
    n = len(gains)
    u = np.zeros(25, dtype=np.int32)
    u[0] = n
    if offsets is not None:
        u[1:1 + n] = offsets
    u[9:9 + n] = np.asarray(gains, dtype=np.float32).view(np.int32)
    u[17:17 + n] = np.asarray(fades, dtype=np.float32).view(np.int32)
    return u


def _snd(fn, out, inputs, uniforms):
    assert out.dtype == np.int16 and out.ndim == 1 and out.flags.c_contiguous
    ptrs = (C.c_void_p * 8)()
    for i, a in enumerate(inputs):
        assert a.dtype == np.int16 and a.shape == out.shape and a.flags.c_contiguous
        ptrs[i] = a.ctypes.data
    u = np.ascontiguousarray(uniforms, dtype=np.int32)
    assert u.size == 25
    return fn(out.ctypes.data, out.size, ptrs, u.ctypes.data)


def snd_s16i_s16i(out, inputs, uniforms):
    """out (int16, interleaved stereo) += the mix of `inputs`, in place"""
    return _snd(lib().orc_snd_s16i_s16i, out, inputs, uniforms)


def clref_snd_s16i_s16i(out, inputs, uniforms):
    l = clref()
    if l is None or not hasattr(l, "clref_run_snd"):
        raise RuntimeError("libclref.so (with snd_s16i_s16i) not built")
    return _snd(l.clref_run_snd, out, inputs, uniforms)


def me_fullsearch(out, ref, cur, block, window, image_size=None):
    """out: (blocks down, blocks across, 4) uint8, written in place; ref / cur: (h, w) uint8 luma planes"""
    u = np.array([block[0], block[1], window[0], window[1]] + list(image_size or (cur.shape[1], cur.shape[0])), dtype=np.int32)
    o, r, c = _planes([out]), _planes([ref]), _planes([cur])
    return lib().orc_me_fullsearch(o, r, c, u.ctypes.data)


def me_cost_table(n=256):
    t = np.zeros(n, dtype=np.float32)
    lib().orc_me_cost_table(t.ctypes.data, n)
    return t
