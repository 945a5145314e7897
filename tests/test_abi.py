"""CPU-side checks of the C-ABI library: it loads, exports every symbol the header
declares, and FAILS LOUDLY without a gfx950 device (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

from spumoni_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    capi.build()
    return capi.lib()


def test_header_symbols_exported(built):
    hdr = open(os.path.join(ROOT, "include", "spumoni_gpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(spx_[a-z_0-9]+)\s*\(", hdr)))
    assert declared, "no declarations found"
    assert sorted(capi.EXPORTS) == declared
    for name in declared:
        assert hasattr(built, name), name


def test_no_gpu_fails_loudly(built):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert built.spx_device_count() == 0
    heads = np.array([0, 65, 67], dtype=np.uint8)
    lens = np.array([1, 2, 3], dtype=np.uint64)
    thr = np.zeros(3, dtype=np.uint64)
    h = built.spx_index_from_runs(
        heads.ctypes.data_as(ctypes.c_void_p), lens.ctypes.data_as(ctypes.c_void_p),
        thr.ctypes.data_as(ctypes.c_void_p), 3, None, None, None, None, 0, 0)
    assert not h
    msg = built.spx_last_error().decode()
    assert "no CPU fallback" in msg


def test_product_does_not_touch_the_oracle():
    """The shipped path must never import / link / include anything under oracle/."""
    pkg = os.path.join(ROOT, "spumoni_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".inc", ".cpp", ".h", ".hpp", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle" not in txt, os.path.join(dp, f)
                assert "fake_device" not in txt and "fake_spumoni" not in txt, os.path.join(dp, f)  # (the CPU tier's stand-in: tests only)


def test_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: include/spumoni_gpu.h compiles as C99 (-pedantic) on its own -- no C++, no torch or
    HIP types in a signature."""
    import subprocess

    src = tmp_path / "h.c"
    src.write_text('#include "spumoni_gpu.h"\nint main(void) { return SPX_OK; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                        "-fsyntax-only", str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    code = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "spumoni_gpu.h")).read(), flags=re.S)  # (comments cite them)
    assert "torch" not in code and "hipStream_t" not in code and "std::" not in code and "#include <hip" not in code


def test_the_fake_device_stays_inside_the_tests():
    """tests/fake_device (the C-ABI answered by the oracle, for the host's CPU tier) is test infrastructure: neither the
    build entry point nor the benchmark knows of it."""
    for f in ("__graft_entry__.py", "bench.py"):
        txt = open(os.path.join(ROOT, f)).read()
        assert "fake_device" not in txt and "fake_spumoni" not in txt, f
