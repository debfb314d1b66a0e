import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session")
def fhe():
    """The product binding.  Builds the HIP library if it is missing (hipcc cross-compiles here)."""
    import learn_fhe_amd as F
    if not os.path.exists(F.lib_path()):
        F.build()
    F.lib()
    return F


@pytest.fixture(scope="session")
def cref():
    from oracle import cref as R
    R.lib()
    return R
