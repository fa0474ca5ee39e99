"""Builds and runs tests/cpp/test_host_mirror.cpp: the C++ mirror of the reference's operator surface
(swiftvideo_amd/host/swiftvideo_hip.hpp) over the C ABI, checked against the oracle."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
EXE = ROOT / "tests" / "cpp" / "test_host_mirror"


def _build(built):
    from oracle import oracle as O
    O.build()
    src = ROOT / "tests" / "cpp" / "test_host_mirror.cpp"
    hdr = ROOT / "swiftvideo_amd" / "host" / "swiftvideo_hip.hpp"
    if EXE.exists() and EXE.stat().st_mtime > max(src.stat().st_mtime, hdr.stat().st_mtime):
        return
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-o", str(EXE), str(src),
           f"-L{ROOT / 'swiftvideo_amd'}", "-lchipvideo", f"-L{ROOT / 'oracle'}", "-loracle",
           "-L/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib",
           f"-Wl,-rpath,{ROOT / 'swiftvideo_amd'}", f"-Wl,-rpath,{ROOT / 'oracle'}", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)


def test_cpp_host_mirror_cpu(built):
    _build(built)
    out = subprocess.run([str(EXE), "cpu"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.gpu
def test_cpp_host_mirror_gpu(built):
    _build(built)
    out = subprocess.run([str(EXE), "gpu"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
