import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
