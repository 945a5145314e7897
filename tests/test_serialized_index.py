"""CPU: reader of the serialised index (<ref>.thrbv.spumoni / .thrbv.ms) in the C++ host against
the stream layout restated in spumoni_amd/csrc/host/index_files.cpp.  The writer lives in
spumoni_amd/sdsl_streams.py (build_index.py --serialized writes with it); both follow the same restatement of sdsl-lite / r-index framing, so this
pins self-consistency only -- the layout itself is UNVERIFIED against an upstream-built file."""
import os
import subprocess

import numpy as np
import pytest

from tests import cases
from tests.sdsl_files import write_thrbv

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_BIN = os.path.join(ROOT, "spumoni_amd", "bin", "spumoni")


def _dump(path, mode):
    out = subprocess.run([HOST_BIN, "dump-index", path, mode], capture_output=True)
    assert out.returncode == 0, out.stderr.decode()
    d = {}
    lines = out.stdout.decode().splitlines()
    _, n, _, r = lines[0].split()
    d["n"], d["r"] = int(n), int(r)
    for ln in lines[1:]:
        tag, *vals = ln.split()
        d[tag] = [int(v) for v in vals]
    return d


@pytest.mark.parametrize("seed,n,letters", [(1, 400, list(b"ACGT")), (2, 3000, list(b"ACGTN")),
                                            (3, 2500, [3, 4, 5, 90, 127, 128, 129, 200, 255]), (4, 64, [ord("A")])])
def test_round_trip(built_all, tmp_path, seed, n, letters):
    raw, _ = cases.real_case(seed, n, letters)
    heads = np.maximum(raw.heads.numpy(), 1)
    for mode, ext in (("P", ".thrbv.spumoni"), ("M", ".thrbv.ms")):
        path = str(tmp_path / ("idx" + ext))
        write_thrbv(path, heads, raw.lens.numpy(), raw.thr.numpy(),
                    raw.ssa.numpy() if mode == "M" else None, raw.esa.numpy() if mode == "M" else None)
        d = _dump(path, mode)
        assert d["n"] == raw.n and d["r"] == raw.r
        assert d["heads"] == heads.tolist()
        assert d["lens"] == raw.lens.tolist()
        assert d["thr"] == raw.thr.tolist()
        if mode == "M":
            assert d["ssa"] == raw.ssa.tolist() and d["esa"] == raw.esa.tolist()


def test_truncated_or_foreign_file_is_an_error(built_all, tmp_path):
    raw, _ = cases.real_case(5, 500, list(b"ACGT"))
    path = str(tmp_path / "idx.thrbv.spumoni")
    write_thrbv(path, np.maximum(raw.heads.numpy(), 1), raw.lens.numpy(), raw.thr.numpy())
    blob = open(path, "rb").read()
    for cut in (10, 2000, len(blob) - 3):
        open(path, "wb").write(blob[:cut])
        out = subprocess.run([HOST_BIN, "dump-index", path, "P"], capture_output=True)
        assert out.returncode == 1 and b"unexpected layout" in out.stderr
    open(path, "wb").write(blob + b"xx")
    out = subprocess.run([HOST_BIN, "dump-index", path, "P"], capture_output=True)
    assert out.returncode == 1


def test_builder_writes_the_serialized_indexes(built_all, tmp_path):
    """`python -m spumoni_amd.build_index --serialized`: <prefix>.thrbv.spumoni / .thrbv.ms beside the raw run files
    (SURVEY f4 lists them; compute_ms_pml.cpp:192-213, 517-542), read back by the host's loader to the raw files' arrays."""
    import sys

    rng = np.random.default_rng(9)
    fa = tmp_path / "ref.fa"
    with open(fa, "w") as f:
        for i in range(3):
            f.write(f">s{i}\n" + "".join(rng.choice(list("ACGT"), size=300)) + "\n")
    out = subprocess.run([sys.executable, "-m", "spumoni_amd.build_index", "-r", str(fa), "-o", str(tmp_path / "idx"), "--serialized"],
                         capture_output=True, cwd=ROOT)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    prefix = str(tmp_path / "idx.fa")

    def five(path):  # r x 5-byte little-endian values (include/ms_rle_string.hpp:246-247, thresholds_ds.hpp:393-417)
        b = np.fromfile(path, dtype=np.uint8).reshape(-1, 5).astype(np.uint64)
        return sum(b[:, i] << np.uint64(8 * i) for i in range(5)).tolist()

    heads = np.maximum(np.fromfile(prefix + ".bwt.heads", dtype=np.uint8), 1).tolist()
    lens, thr = five(prefix + ".bwt.len"), five(prefix + ".thr_pos")
    for mode, ext in (("P", ".thrbv.spumoni"), ("M", ".thrbv.ms")):
        d = _dump(prefix + ext, mode)
        assert d["r"] == len(heads) and d["n"] == sum(lens)
        assert d["heads"] == heads and d["lens"] == lens and d["thr"] == thr, mode
