"""The reference's only compute test, replayed: Tests/swiftVideoInternalTests/computeTests.swift:9-39
(`defaultKernelSearch`): 13 kernel-name strings round-trip through defaultComputeKernelFromString,
with img_clear_rgba resolving to img_clear_bgra."""
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
