"""CPU: the C++ host read batching (spumoni_amd/csrc/host/reads.cpp) against the oracle's
plain-C restatement of BatchLoader (oracle/orc_run.c) on well-formed and malformed inputs."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_BIN = os.path.join(ROOT, "spumoni_amd", "bin", "spumoni")
ORC_RUN = os.path.join(ROOT, "oracle", "orc_run")


@pytest.fixture(scope="module")
def bins():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "spumoni_amd", "csrc"), "-j4"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "spumoni_amd", "csrc", "host")], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    return HOST_BIN, ORC_RUN


def _rand_seq(rng, n):
    return "".join(rng.choice(list("ACGTacgtN"), size=n))


def _cases(rng):
    big_fa = "".join(f">read{i} len={i}\tx\n{_rand_seq(rng, 60)}\n{_rand_seq(rng, int(rng.integers(1, 60)))}  \n" for i in range(120))
    big_fq = "".join(f"@q{i} d\n{_rand_seq(rng, 151)}\n+\n{'I' * 151}\n" for i in range(40))
    return {
        "simple.fa": ">r1\nACGT\n>r2\nGGTT\n",
        "no_trailing_newline.fa": ">r1\nACGT\n>r2\nGGTT",
        "desc_and_multiline.fa": ">r1 some description\nACGT\nacgt\n>r2\tx\nTTTT\n\n>r3\r\nAAAA\r\n",
        "blank_inside.fa": ">r1\nAC\n\nGT\n>r2\nGG\n",
        "many_batches.fa": big_fa,
        "many_batches_nonl.fa": big_fa.rstrip("\n"),
        "empty_seq_last.fa": ">r1\nACGT\n>r2x\n",
        "simple.fq": "@q1\nACGT\n+\nIIII\n@q2 d\nGGTT\n+\nIIII",
        "tail_quirk.fq": "@q1\nACGT\n+\nIIII\n@q2\nGGTT\n+\nIIII\n",  # C16: ends with newline -> batch dropped
        "many.fq": big_fq,
        "many_nonl.fq": big_fq.rstrip("\n"),
        "blank_header.fq": "@q1\nACGT\n+\nIIII\n\n@q2\nGGTT\n+\nIIII",
        "empty.fa": "",
    }


def test_read_batching_matches_oracle(bins, tmp_path):
    host, orc = bins
    rng = np.random.default_rng(3)
    for name, text in _cases(rng).items():
        p = tmp_path / name
        p.write_bytes(text.encode())
        a = subprocess.run([host, "dump-reads", str(p)], capture_output=True)
        b = subprocess.run([orc, "x", str(p), "P", "0", "0", "150", "n", "--dump-reads"], capture_output=True)
        assert a.returncode == 0 and b.returncode == 0, (name, a.stderr, b.stderr)
        assert a.stdout == b.stdout, name
    # sanity on the semantics themselves
    out = subprocess.run([host, "dump-reads", str(tmp_path / "desc_and_multiline.fa")], capture_output=True).stdout.decode()
    assert "r1 \tACGTacgt\n" in out  # id keeps the whitespace char (Appendix C6)
    out = subprocess.run([host, "dump-reads", str(tmp_path / "tail_quirk.fq")], capture_output=True).stdout.decode()
    assert out == ""  # FASTQ tail quirk (Appendix C16)
    out = subprocess.run([host, "dump-reads", str(tmp_path / "simple.fq")], capture_output=True).stdout.decode()
    assert "q1\tACGT\n" in out and "q2 \tGGTT\n" in out
