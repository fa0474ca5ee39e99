"""The C-ABI library loads on a machine without a GPU and exports every symbol that
include/chipvideo.h declares; struct layouts match the header; device-less calls fail with
an error code instead of crashing."""
import ctypes as C
import re
from pathlib import Path

import pytest

from swiftvideo_amd import chipvideo as cv

ROOT = Path(__file__).resolve().parents[1]
HEADER = (ROOT / "include" / "chipvideo.h").read_text()


def declared_functions():
    names = re.findall(r"\b(chv_[a-z0-9_]+)\s*\(", HEADER)
    return sorted(set(names))


def test_every_declared_symbol_is_exported(built):
    lib = C.CDLL(str(cv.LIB_PATH))
    declared = declared_functions()
    assert len(declared) >= 30
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, f"declared in chipvideo.h but not exported: {missing}"


def test_binding_covers_the_header(built):
    assert sorted(cv._SIGNATURES) == declared_functions()
    cv.load()


def test_struct_layouts():
    assert C.sizeof(cv.Uniforms) == 236                      # ImageUniforms, compute.swift:76-86
    assert cv.Uniforms.fill_color.offset == 192 and cv.Uniforms.input_size.offset == 208
    assert cv.Uniforms.output_size.offset == 216 and cv.Uniforms.opacity.offset == 224
    assert cv.Uniforms.image_time.offset == 228 and cv.Uniforms.target_time.offset == 232
    assert C.sizeof(cv.Plane) == 32 and C.sizeof(cv.Image) == 16 + 3 * 32
    assert C.sizeof(cv.KernelOpts) == 16
    assert C.sizeof(cv.SndUniforms) == 100 and cv.SndUniforms.input_gains.offset == 36 and cv.SndUniforms.input_fade.offset == 68     # kernels.cl.swift:536-541
    assert C.sizeof(cv.MeUniforms) == 24 and cv.MeUniforms.image_size.offset == 16                                                       # kernels.metal:33-37
    assert C.sizeof(cv.Layer) == 8 + C.sizeof(cv.Image) + 236 + 16 - 4 or C.sizeof(cv.Layer) % 8 == 0


def test_status_strings_follow_compute_error(built):
    lib = cv.load()
    cases = {0: "success", 1: "invalidValue", 2: "outOfMemory", 3: "invalidContext", 4: "badTarget",
             5: "badInputData", 6: "notImplemented", 7: "computeKernelNotFound", 8: "deviceNotAvailable",
             9: "invalidDevice", 10: "invalidOperation", 11: "badContextState", 12: "invalidPlatform", 13: "unknownError"}
    for code, name in cases.items():
        assert lib.chv_error_string(code).decode() == name
        assert cv.STATUS_NAMES[code] == name
    assert lib.chv_version() == 0x000100


def test_null_and_bad_handles_return_codes_not_crashes(built):
    lib = cv.load()
    assert lib.chv_context_destroy(None) == 3
    assert lib.chv_pass_begin(None) == 3
    assert lib.chv_pass_end(None, 1) == 3
    assert lib.chv_buffer_free(None) == 1
    assert lib.chv_batch_destroy(None) == 1
    assert lib.chv_event_destroy(None) == 1
    assert lib.chv_device_count(None) == 1
    k = C.c_int(0)
    assert lib.chv_kernel_from_string(None, C.byref(k)) == 1
    assert lib.chv_kernel_name(99) is None
    out = C.c_void_p()
    assert lib.chv_buffer_alloc(None, 16, C.byref(out)) == 3 and not out.value


def test_no_device_is_an_error_not_a_fallback(built):
    """On a box without a usable gfx950 device the product refuses to create a context; there is
    no CPU pixel path (ComputeError.deviceNotAvailable, compute.swift:121-129)."""
    from swiftvideo_amd import compute as sv
    if sv.hasAvailableComputeDevices("GPU"):
        pytest.skip("a GPU is present")
    with pytest.raises(sv.ComputeError) as e:
        sv.makeComputeContext(forType="GPU")
    assert e.value.case == "deviceNotAvailable"


def test_header_is_plain_c_and_layouts_match_the_binding(tmp_path):
    """include/chipvideo.h is what a Swift module map (or cgo, JNI, ...) consumes: it must compile as C99 with no
    C++-isms, and the structs the ctypes binding declares must have the C compiler's sizes and offsets."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    src = tmp_path / "layout.c"
    src.write_text('''
#include <stddef.h>
#include <stdio.h>
#include "chipvideo.h"
int main(void) {
    printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(chv_plane), sizeof(chv_image), sizeof(chv_uniforms), sizeof(chv_kernel_opts),
           sizeof(chv_layer), sizeof(chv_tick), sizeof(chv_device_info), sizeof(chv_custom_args));
    printf("%zu %zu %zu %zu %zu\\n", offsetof(chv_plane, pitch), offsetof(chv_image, planes), offsetof(chv_layer, uniforms),
           offsetof(chv_layer, opts), offsetof(chv_tick, layers));
    return 0;
}
''')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", f"-I{ROOT / 'include'}", str(src), "-o", str(exe)])
    sizes, offsets = subprocess.check_output([str(exe)], text=True).strip().splitlines()
    sizes = [int(x) for x in sizes.split()]
    offsets = [int(x) for x in offsets.split()]
    assert sizes[:7] == [C.sizeof(cv.Plane), C.sizeof(cv.Image), C.sizeof(cv.Uniforms), C.sizeof(cv.KernelOpts), C.sizeof(cv.Layer),
                         C.sizeof(cv.Tick), C.sizeof(cv.DeviceInfo)]
    assert sizes[7] == 6 * 80 + 8 + 256                      # chv_custom_args: 6 images, two counts, 256 uniform bytes
    assert offsets == [cv.Plane.pitch.offset, cv.Image.planes.offset, cv.Layer.uniforms.offset, cv.Layer.opts.offset,
                       cv.Tick.layers.offset]


def test_build_flags_report_a_product_build(built):
    """chv_build_flags(): the shipped library is a gfx950 build with every timing-only ablation off (a -DCHV_ABL=n build
    produces wrong pixels by design, profiles/r02_notes.md section 6)."""
    flags = cv.build_flags()
    assert flags.startswith("arch=gfx950;")
    abl = re.findall(r"abl=(\d+)", flags)
    assert len(abl) >= 5 and all(a == "0" for a in abl), flags          # tick_bgra_wave, tick_yuv_wave, tick_bgra_stream, tick_yuv_stream, lanczos3
    assert re.search(r";hipcc=\d+\.\d+\.", flags) and re.search(r";clang=\d+\.\d+;", flags), flags


def test_build_reports_whether_it_compiled(built, capsys):
    """__graft_entry__.build() prints one `chipvideo build_mode=…` line; a second call over an up-to-date tree is "reused"."""
    import __graft_entry__ as entry
    from swiftvideo_amd import build as B
    entry.build()
    entry.build()
    assert B.last_build["mode"] == "reused" and B.last_build["objects_rebuilt"] == [] and not B.last_build["relinked"]
    lines = [l for l in capsys.readouterr().out.splitlines() if l.startswith("chipvideo build_mode=")]
    assert len(lines) == 2 and "build_mode=reused" in lines[-1] and "flags=arch=gfx950;" in lines[-1]


def test_switch_hook_validates_names(built):
    lib = cv.load()
    assert lib.chv_debug_set_switch(b"CHV_NO_SUCH_SWITCH", b"1") == 1
    assert lib.chv_debug_set_switch(None, b"1") == 1
    for name in ("CHV_FORCE_GENERAL", "CHV_BGRA_PATH", "CHV_WAVE_ROWS", "CHV_TILE_ROWS", "CHV_SAME_GEOM", "CHV_DESC", "CHV_STREAM"):
        assert lib.chv_debug_set_switch(name.encode(), b"") == 0
        assert lib.chv_debug_set_switch(name.encode(), None) == 0


def test_counter_hook_validates_names(built):
    """chv_debug_get_counter (the geometry-table store in numbers): unknown names and null arguments are refused; the known ones read 0 in a
    process that has launched nothing (no device needed)"""
    import ctypes as C
    lib = cv.load()
    v = C.c_ulonglong(7)
    assert lib.chv_debug_get_counter(b"no_such_counter", C.byref(v)) == 1
    assert lib.chv_debug_get_counter(None, C.byref(v)) == 1
    assert lib.chv_debug_get_counter(b"geom_store_patched", None) == 1
    for name in ("geom_store_patched", "geom_store_batch_hits", "geom_store_builds", "geom_store_bytes", "geom_store_tables"):
        v.value = 7
        assert lib.chv_debug_get_counter(name.encode(), C.byref(v)) == 0 and v.value == 0, name
        assert cv.get_counter(name) == 0
