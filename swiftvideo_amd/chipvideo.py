"""ctypes binding of the CHIPVideo C ABI (include/chipvideo.h).

This is the Python spelling of the stub a host language writes against the
library (INTEGRATION.md shows the Swift one).  It contains no pixel code and no
fallback: if ``libchipvideo.so`` is missing or a call fails, it raises.
"""
import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "libchipvideo.so"

# ---- status / enums (chipvideo.h) ---------------------------------------------
OK = 0
STATUS_NAMES = {
    0: "success", 1: "invalidValue", 2: "outOfMemory", 3: "invalidContext", 4: "badTarget",
    5: "badInputData", 6: "notImplemented", 7: "computeKernelNotFound", 8: "deviceNotAvailable",
    9: "invalidDevice", 10: "invalidOperation", 11: "badContextState", 12: "invalidPlatform",
    13: "unknownError",
}

K_IMG_NV12_NV12, K_IMG_BGRA_NV12, K_IMG_RGBA_NV12, K_IMG_BGRA_BGRA = 0, 1, 2, 3
K_IMG_Y420P_Y420P, K_IMG_Y420P_NV12, K_IMG_CLEAR_NV12, K_IMG_CLEAR_YUVS = 4, 5, 6, 7
K_IMG_CLEAR_BGRA, K_IMG_CLEAR_Y420P, K_IMG_CLEAR_RGBA, K_IMG_RGBA_Y420P = 8, 9, 10, 11
K_IMG_BGRA_Y420P, K_SND_S16I_S16I, K_ME_FULLSEARCH = 12, 13, 14
K_IMG_NV12_BGRA, K_IMG_Y420P_BGRA, K_IMG_BGRA_BGRA_TX, K_IMG_RGBA_BGRA_TX = 32, 33, 34, 35
K_IMG_BGRA_NV12_INT, K_IMG_RGBA_NV12_INT, K_IMG_BGRA_Y420P_INT, K_IMG_RGBA_Y420P_INT = 36, 37, 38, 39

FMT_NV12, FMT_NV21, FMT_YUVS, FMT_ZVUY, FMT_Y420P, FMT_Y422P, FMT_Y444P, FMT_RGBA, FMT_BGRA = range(9)
FMT_INVALID = 11
CSC_BT601_LIMITED, CSC_BT709_LIMITED, CSC_BT601_FULL, CSC_BT709_FULL = range(4)
MAX_LAYERS = 16


class ComputeError(RuntimeError):
    """Mirror of `enum ComputeError` (compute.swift:22-39): `.case` is the Swift case name."""

    def __init__(self, status, detail=""):
        self.status = int(status)
        self.case = STATUS_NAMES.get(self.status, "unknownError")
        self.detail = detail
        super().__init__(f"ComputeError.{self.case}" + (f": {detail}" if detail else ""))


class DeviceInfo(C.Structure):
    _fields_ = [("index", C.c_int32), ("available", C.c_int32), ("device_type", C.c_int32),
                ("vendor_id", C.c_int32), ("compute_units", C.c_int32), ("supports_images", C.c_int32),
                ("total_memory", C.c_uint64), ("name", C.c_char * 128), ("arch", C.c_char * 32)]


class Plane(C.Structure):
    _fields_ = [("buffer", C.c_void_p), ("offset", C.c_size_t), ("width", C.c_int32),
                ("height", C.c_int32), ("pitch", C.c_int32), ("components", C.c_int32)]


class Image(C.Structure):
    _fields_ = [("format", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("n_planes", C.c_int32), ("planes", Plane * 3)]


class Uniforms(C.Structure):
    _fields_ = [("transform", C.c_float * 16), ("texture_transform", C.c_float * 16),
                ("border_matrix", C.c_float * 16), ("fill_color", C.c_float * 4),
                ("input_size", C.c_float * 2), ("output_size", C.c_float * 2),
                ("opacity", C.c_float), ("image_time", C.c_float), ("target_time", C.c_float)]


class KernelOpts(C.Structure):
    _fields_ = [("colorspace", C.c_int32), ("reserved", C.c_int32 * 3)]


class Layer(C.Structure):
    _fields_ = [("kernel", C.c_int32), ("image", Image), ("uniforms", Uniforms), ("opts", KernelOpts)]


class Tick(C.Structure):
    _fields_ = [("target", Image), ("clear_first", C.c_int32), ("n_layers", C.c_int32),
                ("layers", C.POINTER(Layer))]


class SndUniforms(C.Structure):      # BufferUniforms, kernels.cl.swift:536-541
    _fields_ = [("input_count", C.c_int32), ("input_offsets", C.c_int32 * 8), ("input_gains", C.c_float * 8), ("input_fade", C.c_float * 8)]


class MeUniforms(C.Structure):       # MotionEstimationUniforms, kernels.metal:33-37
    _fields_ = [("block_size", C.c_int32 * 2), ("search_window_size", C.c_int32 * 2), ("image_size", C.c_int32 * 2)]


assert C.sizeof(Uniforms) == 236 and C.sizeof(SndUniforms) == 100 and C.sizeof(MeUniforms) == 24

_SIGNATURES = {
    # name: (restype, argtypes) — one entry per function declared in include/chipvideo.h
    "chv_error_string": (C.c_char_p, [C.c_int]),
    "chv_last_error_detail": (C.c_char_p, []),
    "chv_version": (C.c_int, []),
    "chv_build_flags": (C.c_char_p, []),
    "chv_debug_set_switch": (C.c_int, [C.c_char_p, C.c_char_p]),
    "chv_debug_get_counter": (C.c_int, [C.c_char_p, C.POINTER(C.c_ulonglong)]),
    "chv_kernel_from_string": (C.c_int, [C.c_char_p, C.POINTER(C.c_int)]),
    "chv_kernel_name": (C.c_char_p, [C.c_int]),
    "chv_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "chv_device_info_get": (C.c_int, [C.c_int, C.POINTER(DeviceInfo)]),
    "chv_context_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "chv_context_share": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "chv_context_destroy": (C.c_int, [C.c_void_p]),
    "chv_context_device": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "chv_context_stream": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "chv_context_numa_node": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "chv_buffer_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "chv_buffer_wrap": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "chv_buffer_free": (C.c_int, [C.c_void_p]),
    "chv_buffer_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "chv_plane_alloc": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "chv_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]),
    "chv_host_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "chv_host_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "chv_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t]),
    "chv_download_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t]),
    "chv_pass_begin": (C.c_int, [C.c_void_p]),
    "chv_run_kernel": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(Image), C.POINTER(Image), C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(KernelOpts)]),
    "chv_pass_end": (C.c_int, [C.c_void_p, C.c_int]),
    "chv_composite": (C.c_int, [C.c_void_p, C.POINTER(Image), C.c_int, C.POINTER(Layer), C.c_int]),
    "chv_batch_create": (C.c_int, [C.c_void_p, C.POINTER(Tick), C.c_int, C.POINTER(C.c_void_p)]),
    "chv_batch_run": (C.c_int, [C.c_void_p, C.c_void_p]),
    "chv_batch_destroy": (C.c_int, [C.c_void_p]),
    "chv_batch_describe": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]),
    "chv_scale_lanczos": (C.c_int, [C.c_void_p, C.POINTER(Image), C.POINTER(Image)]),
    "chv_scale_lanczos_batch": (C.c_int, [C.c_void_p, C.POINTER(Image), C.POINTER(Image), C.c_int]),
    "chv_custom_prelude": (C.c_char_p, []),
    "chv_kernel_build": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p]),
    "chv_run_custom": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(Image), C.POINTER(Image), C.c_int, C.c_void_p,
                                 C.c_size_t, C.c_int]),
    "chv_event_create": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "chv_event_record": (C.c_int, [C.c_void_p, C.c_void_p]),
    "chv_event_wait": (C.c_int, [C.c_void_p, C.c_void_p]),
    "chv_event_synchronize": (C.c_int, [C.c_void_p]),
    "chv_event_elapsed_ms": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]),
    "chv_event_destroy": (C.c_int, [C.c_void_p]),
    "chv_device_synchronize": (C.c_int, [C.c_void_p]),
}

_lib = None


def load():
    """Load libchipvideo.so (built in-tree by swiftvideo_amd/build.py). Raises if absent."""
    global _lib
    if _lib is None:
        # CHV_LIB: another build of the same library (A/B measurements, tools/build_variant.sh)
        path = LIB_PATH
        if os.environ.get("CHV_LIB"):
            path = Path(os.environ["CHV_LIB"])
            path = path if path.is_absolute() else _HERE.parent / path      # relative to the repository root
        if not path.exists():
            raise ImportError(
                f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback for the picture kernels)")
        lib = C.CDLL(str(path), mode=getattr(os, "RTLD_NOW", 2))
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(status):
    if status != OK:
        detail = load().chv_last_error_detail()
        raise ComputeError(status, detail.decode() if detail else "")


def build_flags():
    """chv_build_flags(): what the loaded library was built with (arch, ablation switches)."""
    return load().chv_build_flags().decode()


def get_counter(name):
    """chv_debug_get_counter: the device's geometry-table store in numbers (tests / probes)."""
    v = C.c_ulonglong(0)
    check(load().chv_debug_get_counter(name.encode(), C.byref(v)))
    return v.value


def set_switch(name, value):
    """chv_debug_set_switch: measurement / test hook for the path-selection switches (None restores the default)."""
    check(load().chv_debug_set_switch(name.encode(), None if value is None else str(value).encode()))


def kernel_from_string(name):
    k = C.c_int(-1)
    check(load().chv_kernel_from_string(name.encode(), C.byref(k)))
    return k.value


def kernel_name(kernel):
    s = load().chv_kernel_name(int(kernel))
    return s.decode() if s else None


def device_count():
    n = C.c_int(0)
    check(load().chv_device_count(C.byref(n)))
    return n.value


def device_info(index):
    info = DeviceInfo()
    check(load().chv_device_info_get(index, C.byref(info)))
    return info
