import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle

    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def built_all():
    """Make sure every native artefact the tests run exists (make is a no-op when up to date)."""
    import subprocess

    for d in ("spumoni_amd/csrc", "spumoni_amd/csrc/host", "oracle"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, d), "-j4"], stdout=subprocess.DEVNULL)
    return True
