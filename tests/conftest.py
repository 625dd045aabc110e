import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ska-sdp-accelerate-gridding_amd")
for p in (ROOT, os.path.join(PKG, "python")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Built artefacts are git-ignored: a fresh checkout has no .so files.  Build them once (hipcc
    cross-compiles gfx950 without a GPU; gcc for the oracle) so the suite is self-contained."""
    needed = [os.path.join(PKG, "lib", "libgridhip.so"), os.path.join(PKG, "lib", "libgridhip_io.so"),
              os.path.join(ROOT, "oracle", "libgridref.so")]
    if not all(os.path.exists(p) for p in needed):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def oracle():
    """The C oracle (oracle/libgridref.so), built on demand with gcc."""
    from oracle import gridref_c
    gridref_c.lib()
    return gridref_c


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load


@pytest.fixture(scope="session")
def ctx():
    """A gridhip context on device 0; fails loudly if the HIP library or the GPU is missing."""
    import gridhip
    c = gridhip.Context(0)
    yield c
    c.close()
