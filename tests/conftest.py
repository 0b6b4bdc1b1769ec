import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_sessionstart(session):
    """The HIP library is built in-tree and git-ignored: a fresh checkout has none.  Build it once (hipcc cross-compiles
    gfx950 without a GPU) so that the C-ABI tests and the GPU tests find it; a box without hipcc keeps the clear error of
    `_lib.load()`."""
    from open_o3_video_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        try:
            build.find_hipcc()
        except RuntimeError:
            return
        build.build(verbose=False)
