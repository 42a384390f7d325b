import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as f:
        return {k: f[k] for k in f.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The C half of the oracle is a build product (oracle/_build); make it."""
    import subprocess
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True,
                   stdout=subprocess.DEVNULL)
