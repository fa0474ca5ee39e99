"""In-tree build of libchipvideo.so (hipcc, gfx950).  No JIT cache: the .so sits next
to this file so that it travels to the GPU box with the repository snapshot."""
import os
import subprocess
from pathlib import Path

HERE = Path(__file__).resolve().parent


def build(verbose=False, jobs=None):
    jobs = jobs or min(8, os.cpu_count() or 2)
    cmd = ["make", "-C", str(HERE / "csrc"), f"-j{jobs}"]
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(cmd, stdout=out)
    lib = HERE / "libchipvideo.so"
    if not lib.exists():
        raise RuntimeError("libchipvideo.so was not produced")
    return lib


if __name__ == "__main__":
    print(build(verbose=True))
