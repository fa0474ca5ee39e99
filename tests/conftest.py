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
