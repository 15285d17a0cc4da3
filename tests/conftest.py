import os
import sys
from pathlib import Path

# The dense oracle factorises 400 .. 1500-row matrices a few thousand times per suite.  On a box that shows a hundred cores to a
# container allowed sixteen, OpenBLAS starts a thread per visible core and a 30 ms factorisation takes 1.3 s: the suite spent most of
# its time there.  Four threads (set before numpy loads where possible, and again through threadpoolctl below for the case where a
# plugin has imported numpy first).
os.environ.setdefault("OMP_NUM_THREADS", "4")
os.environ.setdefault("OPENBLAS_NUM_THREADS", "4")

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _blas_threads():
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:
        yield
        return
    with threadpool_limits(limits=4):
        yield


@pytest.fixture(scope="session")
def pkg():
    from __graft_entry__ import load_package

    return load_package()


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        d = np.load(ROOT / "tests" / "golden" / f"{name}.npz")
        return {k: d[k] for k in d.files}

    return load


def scaled_err(a, b, scale):
    import numpy as np

    return float(np.abs((np.asarray(a) - np.asarray(b)) / np.asarray(scale).reshape(-1, 1, 1)).max())
