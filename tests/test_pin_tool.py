"""CPU: tools/pin_upstream.py on a synthetic directory.  The "upstream" outputs here are the oracle's own
(there is no upstream build offline), so this checks the tool's mechanics -- what it reads, what it
compares, that a difference anywhere is a FAIL -- not parity.  The layout of the directory is the
contract an external upstream run has to fill (tools/pin_upstream.py's docstring)."""
import os
import shutil
import subprocess
import sys

import numpy as np

from tests import cases
from tests.sdsl_files import write_doc_array, write_null_db, write_thrbv

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "pin_upstream.py")
ORC_RUN = os.path.join(ROOT, "oracle", "orc_run")


def _make_dir(tmp_path):
    d = tmp_path / "upstream"
    d.mkdir()
    raw, text = cases.real_case(17, 5000, list(b"ACGT"), ndocs=3)
    prefix = str(d / "ref.fa")
    open(prefix, "w").write(">dummy\n")
    raw.write_raw_files(prefix)
    text.tofile(prefix + ".rawtext")
    heads = np.maximum(raw.heads.numpy(), 1)
    write_thrbv(prefix + ".thrbv.spumoni", heads, raw.lens.numpy(), raw.thr.numpy())
    write_thrbv(prefix + ".thrbv.ms", heads, raw.lens.numpy(), raw.thr.numpy(), raw.ssa.numpy(), raw.esa.numpy())
    write_doc_array(prefix + ".doc", raw.doc_start.numpy(), raw.doc_end.numpy())
    write_null_db(prefix + ".pmlnulldb", 4.0, [1, 2, 3, 4, 4])
    write_null_db(prefix + ".msnulldb", 9.0, [5, 9, 9, 9])
    rng = np.random.default_rng(3)
    seqs, offs = cases.reads_mixed(rng, text, list(b"ACGT"), 60, 300, [ord("N")])
    with open(d / "reads.fa", "w") as f:
        for q in range(offs.size - 1):
            s = seqs[offs[q]:offs[q + 1]].tobytes().decode()
            if s:
                f.write(f">r{q} desc\n{s}\n")
    (d / "run_flags.txt").write_text("-n\n")
    for mode in ("P", "M"):
        out = d / "expected" / mode
        out.mkdir(parents=True)
        shutil.copy(d / "reads.fa", out / "reads.fa")
        cmd = [ORC_RUN, prefix, str(out / "reads.fa"), mode, "1", "1", "150", "n"] + ([prefix + ".rawtext"] if mode == "M" else [])
        assert subprocess.run(cmd, capture_output=True).returncode == 0
        os.remove(out / "reads.fa")
    return d


def test_pin_tool_passes_and_fails(built_all, tmp_path):
    d = _make_dir(tmp_path)
    r = subprocess.run([sys.executable, TOOL, str(d)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("[PASS]") == 4 and "[FAIL]" not in r.stdout  # 2 reader checks, 2 oracle runs
    # one differing byte in one expected file is a failure
    p = d / "expected" / "P" / "reads.fa.pseudo_lengths"
    blob = bytearray(p.read_bytes())
    i = blob.index(b" ")
    blob[i - 1] = ord("9") if blob[i - 1] != ord("9") else ord("8")
    p.write_bytes(bytes(blob))
    r = subprocess.run([sys.executable, TOOL, str(d)], capture_output=True, text=True)
    assert r.returncode == 1 and "reads.fa.pseudo_lengths" in r.stdout
    # a raw file that disagrees with the serialised index is a failure of check 1
    lens = bytearray(open(d / "ref.fa.bwt.len", "rb").read())
    lens[0] ^= 1
    open(d / "ref.fa.bwt.len", "wb").write(bytes(lens))
    r = subprocess.run([sys.executable, TOOL, str(d)], capture_output=True, text=True)
    assert r.returncode == 1 and "differs in lens" in r.stdout


def test_pin_tool_cli_leg_on_the_fake_device(built_all, fake_device, tmp_path):
    """--gpu (check 3: the `spumoni run` binary on the same directory) goes through its mechanics on the CPU with
    tests/fake_device behind the binary: six PASS lines (2 reader checks, 2 oracle runs, 2 CLI runs)."""
    d = _make_dir(tmp_path)
    env = dict(os.environ, LD_LIBRARY_PATH=fake_device + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""), SPUMONI_CACHE="off")
    r = subprocess.run([sys.executable, TOOL, str(d), "--gpu"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("[PASS]") == 6 and "[FAIL]" not in r.stdout, r.stdout
