import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_sessionstart(session):
    """Build the in-tree native pieces if they are missing (fresh checkout: *.so files are git-ignored).  hipcc
    cross-compiles gfx950 without a GPU; if it is absent the library tests fail loudly, which is the point."""
    try:
        import sparse_rx
        sparse_rx.build_library(force=False)
    except Exception as e:  # pragma: no cover
        print(f"[conftest] could not build libsparse_rx.so: {e}")
