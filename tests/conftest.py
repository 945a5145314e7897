import os
import subprocess
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


@pytest.fixture(scope="session")
def fake_device(tmp_path_factory, built_all):
    """tests/fake_device/fake_spumoni_gpu.c + the oracle sources (no OpenMP: one thread per call, like one device
    queue) as libspumoni_gpu.so in a temporary directory; the host binaries (plain, ASan, TSan) built."""
    d = tmp_path_factory.mktemp("fake_device")
    subprocess.check_call(["gcc", "-O1", "-g", "-std=c11", "-fsigned-char", "-fPIC", "-Wall", "-Wextra", "-Wno-unknown-pragmas", "-shared",
                           "-pthread", "-o", str(d / "libspumoni_gpu.so"), os.path.join(ROOT, "tests", "fake_device", "fake_spumoni_gpu.c"),
                           os.path.join(ROOT, "oracle", "spumoni_oracle.c"), os.path.join(ROOT, "oracle", "orc_digest.c")])
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "spumoni_amd", "csrc", "host"), "all", "san", "-j2"], stdout=subprocess.DEVNULL)
    return str(d)
