import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Build libafx.so in-tree when it is missing or older than its sources (hipcc cross-compiles gfx950 without a GPU)."""
    try:
        from nerf_for_angiography_amd import build as afx_build
        afx_build.build(force=False)
    except Exception as exc:          # the library-load test then fails loudly with the reason
        print(f"[conftest] could not build libafx.so: {exc}")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    return load


def rel_l2(a, b):
    import numpy as np
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / (den if den > 0 else 1.0))
