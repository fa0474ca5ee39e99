import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Make sure the native pieces exist (no-op when already built)."""
    lib = ROOT / "swiftvideo_amd" / "libchipvideo.so"
    if not lib.exists():
        import __graft_entry__ as g
        g.build()
    return lib


@pytest.fixture(scope="session")
def ctx(built):
    """A compute context on device 0.  GPU tests fail (not skip) without a usable device:
    the product has no CPU path to fall back to."""
    from swiftvideo_amd import compute as sv
    c = sv.makeComputeContext(forType="GPU")
    yield c
    sv.destroyComputeContext(c)


@pytest.fixture(scope="session", autouse=True)
def eager_geometry_tables(request):
    """-m gpu sessions: batches build their geometry tables at their FIRST launch (CHV_GEOM_CACHE=eager; the product builds them at the second, so
    that a batch run once never pays) — most batches of this suite run once, and every fuzzer that runs one is meant to reach the table-reading
    instantiations of the strip kernels (eager also lets a transient launch build at a scene's first sighting: the device's store,
    tests/test_gpu_geom_store.py).  The kernels that compute their geometry in place are what every first sighting takes in the product, and
    tests/test_gpu_geom_cache.py forces them onto batches."""
    lib = ROOT / "swiftvideo_amd" / "libchipvideo.so"
    if lib.exists() and os.environ.get("CHV_GEOM_CACHE") is None and "gpu" in (request.config.getoption("-m") or "") and "not gpu" not in (request.config.getoption("-m") or ""):
        from swiftvideo_amd import chipvideo
        chipvideo.set_switch("CHV_GEOM_CACHE", "eager")
    yield


@pytest.fixture
def switch(built):
    """Path-selection switches through the library's own hook (chv_debug_set_switch) instead of the environment: the library
    reads the environment once per process.  Everything set through the fixture goes back to its default afterwards."""
    from swiftvideo_amd import chipvideo
    touched = []

    def _set(name, value):
        chipvideo.set_switch(name, value)
        touched.append(name)

    yield _set
    for name in touched:
        # (CHV_GEOM_CACHE goes back to what the session runs with: eager_geometry_tables)
        chipvideo.set_switch(name, "eager" if name == "CHV_GEOM_CACHE" and os.environ.get("CHV_GEOM_CACHE") is None else None)
