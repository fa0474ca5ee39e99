"""In-tree build of libchipvideo.so (hipcc, gfx950).  No JIT cache: the .so sits next
to this file so that it travels to the GPU box with the repository snapshot.

`build()` says what it did (`last_build`: mode "compiled" when make had to run hipcc for at least one object, "reused" when
every object and the library were newer than their sources) so that a driver can tell a cross-compiled, travelled binary
from one rebuilt on the box."""
import os
import subprocess
import time
from pathlib import Path

HERE = Path(__file__).resolve().parent
last_build = None        # {"mode": "compiled"|"reused", "objects_rebuilt": [...], "seconds": s, "lib": path}


def _object_mtimes():
    return {p.name: p.stat().st_mtime_ns for p in (HERE / "csrc").glob("*.o")}


def build(verbose=False, jobs=None):
    global last_build
    jobs = jobs or min(8, os.cpu_count() or 2)
    lib = HERE / "libchipvideo.so"
    before, lib_before = _object_mtimes(), (lib.stat().st_mtime_ns if lib.exists() else None)
    t0 = time.time()
    cmd = ["make", "-C", str(HERE / "csrc"), f"-j{jobs}"]
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(cmd, stdout=out)
    if not lib.exists():
        raise RuntimeError("libchipvideo.so was not produced")
    after = _object_mtimes()
    rebuilt = sorted(n for n, t in after.items() if before.get(n) != t)
    relinked = lib.stat().st_mtime_ns != lib_before
    last_build = {"mode": "compiled" if (rebuilt or relinked) else "reused", "objects_rebuilt": rebuilt, "relinked": relinked,
                  "seconds": round(time.time() - t0, 2), "lib": str(lib)}
    return lib


if __name__ == "__main__":
    print(build(verbose=True))
    print(last_build)
