"""The reference's only compute test, replayed: Tests/swiftVideoInternalTests/computeTests.swift:9-39
(`defaultKernelSearch`): 13 kernel-name strings round-trip through defaultComputeKernelFromString,
with img_clear_rgba resolving to img_clear_bgra."""
import numpy as np
import pytest

from oracle import oracle as O
from swiftvideo_amd import chipvideo as cv
from swiftvideo_amd import compute as sv

REFERENCE_NAMES = [
    "img_nv12_nv12", "img_bgra_nv12", "img_rgba_nv12", "img_bgra_bgra", "img_y420p_y420p",
    "img_y420p_nv12", "img_clear_nv12", "img_clear_yuvs", "img_clear_bgra", "img_clear_rgba",
    "img_rgba_y420p", "img_bgra_y420p", "img_clear_y420p",
]


def test_default_kernel_search(built):
    for name in REFERENCE_NAMES:
        result = sv.defaultComputeKernelFromString(name)
        if name != "img_clear_rgba":
            assert str(result) == name
        else:
            assert str(result) == "img_clear_bgra"   # compute.swift:101


def test_unknown_name_throws_invalid_value(built):
    for name in ["img_nv21_nv12", "", "img_clear", "IMG_NV12_NV12", "img_nv12_nv12 "]:
        with pytest.raises(sv.ComputeError) as e:
            sv.defaultComputeKernelFromString(name)
        assert e.value.case == "invalidValue"          # compute.swift:106-108


def test_names_findKernel_can_synthesise_for_bgra_canvases(built):
    # mix.video.swift:142-146 builds "img_<input fmt>_<target fmt>"; the reference has no entry for these
    for name in ["img_nv12_bgra", "img_y420p_bgra", "img_bgra_bgra_tx", "img_rgba_bgra_tx"]:
        assert str(sv.defaultComputeKernelFromString(name)) == name


def test_integer_rgb_to_yuv_names(built):
    # the encoder-side kernels: named like the reference's float family with an `_int` suffix
    for name in ["img_bgra_nv12_int", "img_rgba_nv12_int", "img_bgra_y420p_int", "img_rgba_y420p_int"]:
        assert str(sv.defaultComputeKernelFromString(name)) == name
        assert cv.kernel_name(int(sv.ComputeKernel[name])) == name


def test_the_two_idle_kernels_resolve_by_name(built):
    # compute.swift:67,70 declares them, the reference's table (:91-105) leaves them out; here every case with a kernel has a name
    for name in ["snd_s16i_s16i", "me_fullsearch"]:
        assert str(sv.defaultComputeKernelFromString(name)) == name


def test_enum_values_follow_reference_declaration_order(built):
    # compute.swift:49-74
    order = ["img_nv12_nv12", "img_bgra_nv12", "img_rgba_nv12", "img_bgra_bgra", "img_y420p_y420p", "img_y420p_nv12",
             "img_clear_nv12", "img_clear_yuvs", "img_clear_bgra", "img_clear_y420p", "img_clear_rgba",
             "img_rgba_y420p", "img_bgra_y420p", "snd_s16i_s16i", "me_fullsearch"]
    for i, name in enumerate(order):
        assert cv.kernel_name(i) == name
        assert int(sv.ComputeKernel[name]) == i


def test_oracle_and_library_agree_on_ids(built):
    for name, kid in O.KERNEL_IDS.items():
        assert cv.kernel_name(kid) == name


# ---- VideoMixer.findKernel's name synthesis for every (layer format, canvas format) pair -------------------------
# mix.video.swift:142-146: name = "img_" + lowercased(String(describing: layer format) | "clear") + "_" + lowercased(canvas
# format), resolved by defaultComputeKernelFromString — compute.swift:90-110 plus the entries of the INTEGRATION.md hunk.
# Layer formats: every case of `enum PixelFormat` (sample.pict.swift:20-33); canvases: the three the kernels write.
PIXEL_FORMATS = ["nv12", "nv21", "yuvs", "zvuy", "y420p", "y422p", "y444p", "rgba", "bgra", "invalid"]
DOCUMENTED = {
    # canvas: {layer format: kernel the name resolves to}; everything else throws ComputeError.invalidValue, as in the reference
    "nv12": {"clear": "img_clear_nv12", "nv12": "img_nv12_nv12", "y420p": "img_y420p_nv12", "bgra": "img_bgra_nv12", "rgba": "img_rgba_nv12"},
    "y420p": {"clear": "img_clear_y420p", "y420p": "img_y420p_y420p", "bgra": "img_bgra_y420p", "rgba": "img_rgba_y420p"},
    "bgra": {"clear": "img_clear_bgra", "nv12": "img_nv12_bgra", "y420p": "img_y420p_bgra",
             "bgra": "img_bgra_bgra",            # Metal semantics (kernels.metal:52-62): what the unchanged Swift VideoMixer gets
             "rgba": "img_rgba_bgra_tx"},        # "img_rgba_bgra": no reference kernel of that name -> the transform-aware one
}


@pytest.mark.parametrize("canvas", list(DOCUMENTED))
def test_find_kernel_name_synthesis_through_the_c_abi(built, canvas):
    import ctypes as C
    lib = cv.load()
    for layer in ["clear"] + PIXEL_FORMATS:
        name = f"img_{layer}_{canvas}"
        k = C.c_int(-1)
        rc = lib.chv_kernel_from_string(name.encode(), C.byref(k))
        want = DOCUMENTED[canvas].get(layer)
        if want is None:
            assert rc == 1, f"{name}: expected invalidValue, got status {rc} / kernel {k.value}"      # CHV_ERR_INVALID_VALUE
        else:
            assert rc == 0 and lib.chv_kernel_name(k.value).decode() == want, name


def test_mirrors_follow_the_documented_outcome(built):
    """the Python VideoMixer.findKernel (and PictureFilter's) produce the same names; "tx" = the optional mix.video.swift hunk"""
    class M(sv.VideoMixer):
        def __init__(self, family):
            self.bgraKernelFamily = family
    fmt = {"nv12": sv.PixelFormat.nv12, "y420p": sv.PixelFormat.y420p, "bgra": sv.PixelFormat.BGRA, "rgba": sv.PixelFormat.RGBA,
           "nv21": sv.PixelFormat.nv21, "yuvs": sv.PixelFormat.yuvs, "zvuy": sv.PixelFormat.zvuy, "y422p": sv.PixelFormat.y422p,
           "y444p": sv.PixelFormat.y444p, "invalid": sv.PixelFormat.invalid}
    for canvas, table in DOCUMENTED.items():
        target = sv.createPictureSample((8, 8), fmt[canvas])
        for layer in PIXEL_FORMATS:
            image = sv.PictureSample(sv.ImageBuffer(fmt[layer], "cpu", (8, 8), buffers=[np.zeros((8, 32), dtype=np.uint8)]))
            want = table.get(layer)
            if want is None:
                with pytest.raises(sv.ComputeError) as e:
                    M("reference").findKernel(image, target)
                assert e.value.case == "invalidValue"
            else:
                assert str(M("reference").findKernel(image, target)) == want
        assert str(M("reference").findKernel(None, target)) == table["clear"]
    bg = sv.createPictureSample((8, 8), sv.PixelFormat.BGRA)
    assert str(M("tx").findKernel(bg, bg)) == "img_bgra_bgra_tx"
    assert str(M("tx").findKernel(sv.createPictureSample((8, 8), sv.PixelFormat.RGBA), bg)) == "img_rgba_bgra_tx"
