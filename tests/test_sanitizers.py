"""The host side of libchipvideo (swiftvideo_amd/csrc/chipvideo.cpp: contexts, buffers, the descriptor ring, upload / download ordering, batches,
error paths) compiled for the CPU against a stand-in HIP runtime whose streams execute LAZILY (tests/stubhip/), and driven by
tests/stubhip/abi_stress.cpp — single-threaded sequences incl. ring wrap and injected launch failures, then eight threads with contexts of
their own, cross-thread buffer frees and upload / mixer / download contexts — under AddressSanitizer + UBSan and under ThreadSanitizer.
SURVEY 8b's threading and error-convention rows (compute.cuda.swift:308-319, compute.swift:189-190) are what this holds the C ABI to."""
import os
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
STUB = ROOT / "tests" / "stubhip"
OUT = STUB / "_build"


def _build(kind):
    OUT.mkdir(exist_ok=True)
    exe = OUT / f"abi_stress_{kind}"
    srcs = [ROOT / "swiftvideo_amd" / "csrc" / "chipvideo.cpp", ROOT / "include" / "chipvideo.h", ROOT / "swiftvideo_amd" / "csrc" / "device_types.h", ROOT / "swiftvideo_amd" / "csrc" / "geom_cache.h",
            STUB / "stub_runtime.cpp", STUB / "stub_launchers.cpp", STUB / "abi_stress.cpp", STUB / "hip" / "hip_runtime.h", STUB / "build.sh"]
    if not exe.exists() or exe.stat().st_mtime < max(s.stat().st_mtime for s in srcs):
        subprocess.check_call(["bash", str(STUB / "build.sh"), kind, str(exe)])
    return exe


@pytest.mark.parametrize("kind", ["address", "thread"])
def test_c_abi_host_logic_under_sanitizers(kind):
    exe = _build(kind)
    env = dict(os.environ, STUBHIP_DEVICES="2", ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               TSAN_OPTIONS="halt_on_error=0:second_deadlock_stack=1")
    for k in list(env):
        if k.startswith("CHV_"):
            del env[k]
    out = subprocess.run([str(exe), "8"], capture_output=True, text=True, env=env, timeout=900)
    text = out.stdout + out.stderr
    assert out.returncode == 0 and "abi_stress: ok" in text, text[-4000:]
    assert "Sanitizer" not in text and "runtime error" not in text, text[-4000:]


def test_stub_device_refuses_other_architectures():
    """the product's rule (no CPU pixel path, gfx950 only) holds in the stand-in too: a device that reports another architecture is refused"""
    exe = _build("address")
    env = dict(os.environ, STUBHIP_DEVICES="2", STUBHIP_ARCH="gfx90a:sramecc+:xnack-")
    out = subprocess.run([str(exe), "1"], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode != 0 and "deviceNotAvailable" in (out.stdout + out.stderr), (out.stdout + out.stderr)[-2000:]
