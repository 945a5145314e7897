#!/usr/bin/env python3
"""Generates the committed FILE-level fixtures (run from the repo root: python tests/golden/make_golden_files.py).

SURVEY.md 8(c)(4): for a handful of input shapes -- multi-line FASTA with descriptions in the headers and lower-case
stretches, FASTQ content, reads with N / absent letters, a promoted alphabet with bytes >= 128 -- the index files `run`
reads, a reads file, and the output files (.pseudo_lengths / .lengths / .pointers / .doc_numbers / .report) the CPU
oracle harness (oracle/orc_run: classify_reads_pml / classify_reads_ms, BatchLoader and the writers restated,
compute_ms_pml.cpp:845-1217, batch_loader.cpp:26-131) writes for them.  These fixtures are OURS (the reference ships
none and cannot be built offline): they freeze the harness's byte format, so that a later change of the oracle harness
or -- through tests/test_gpu_cli.py, which holds the HIP-backed `spumoni run` against orc_run file by file -- of the
product shows up as a diff against committed data.
"""
import os
import shutil
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from tests import cases  # noqa: E402
from tests.sdsl_files import write_doc_array, write_null_db  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "files")
ORC_RUN = os.path.join(ROOT, "oracle", "orc_run")

# name: (seed, text length, letters, extra read letters, FASTQ?, reads, longest read)
# (FASTQ: BatchLoader::loadBatch returns false when getline meets the end of the file inside a batch
# (batch_loader.cpp:50), so the last, incomplete 1000-base batch of a FASTQ file that ends in a newline is never
# processed -- the fixture is long enough for whole batches and freezes that its tail is dropped)
CASES = {
    "dna_multiline_fasta": (201, 1200, list(b"ACGT"), [ord("N")], False, 24, 70),
    "dna_fastq": (202, 1000, list(b"ACGT"), [ord("N"), ord("Z")], True, 40, 120),
    "promoted_alphabet_fasta": (203, 1500, [3, 4, 5, 60, 127, 128, 129, 200, 255], [250], False, 24, 70),
}
# (mode letter, doc, report, bin width): what is run for every case (the CLI refuses a bin width outside [50, 400])
RUNS = [("P", 1, 1, 50), ("M", 1, 1, 60)]


def write_reads(path, seqs, offs, rng, fastq, printable):
    with open(path, "wb") as f:
        for q in range(offs.size - 1):
            s = seqs[offs[q]: offs[q + 1]].tobytes()
            if not s:
                continue
            if printable and rng.random() < 0.3:
                s = s.lower()
            name = b"read_%d" % q + (b" some description" if q % 4 == 0 else b"")
            if fastq:
                f.write(b"@" + name + b"\n" + s + b"\n+\n" + b"I" * len(s) + b"\n")
            else:
                cut = int(rng.integers(1, len(s) + 1))
                f.write(b">" + name + b"\n" + s[:cut] + b"\n" + (s[cut:] + b"\n" if cut < len(s) else b""))


def outputs(mode, doc, rep):
    exts = [".pseudo_lengths"] if mode == "P" else [".lengths", ".pointers"]
    return exts + ([".doc_numbers"] if doc else []) + ([".report"] if rep else [])


def main():
    shutil.rmtree(OUT, ignore_errors=True)
    for name, (seed, n, letters, extra, fastq, nreads, longest) in CASES.items():
        d = os.path.join(OUT, name)
        os.makedirs(d)
        raw, text = cases.real_case(seed, n, letters, ndocs=3)
        prefix = os.path.join(d, "ref.fa")
        open(prefix, "w").write(">dummy\n")
        raw.write_raw_files(prefix)
        text.tofile(prefix + ".rawtext")
        write_doc_array(prefix + ".doc", raw.doc_start.numpy(), raw.doc_end.numpy())
        write_null_db(prefix + ".pmlnulldb", 4.0, [1, 2, 3, 4, 4, 4, 4, 4])
        write_null_db(prefix + ".msnulldb", 9.0, [5, 9, 9, 9, 9, 9])
        rng = np.random.default_rng(seed + 1000)
        seqs, offs = cases.reads_mixed(rng, text, letters, nreads, longest, extra)
        printable = max(letters) < 128 and min(letters) >= 32
        if not printable:  # (bytes that would end a line or start a record stay out of the reads file)
            seqs = seqs.copy()
            seqs[np.isin(seqs, [10, 13, 62, 64])] = letters[0]
        reads = os.path.join(d, "reads.fa")  # (validate() insists on a .fa name even for FASTQ content)
        write_reads(reads, seqs, offs, np.random.default_rng(seed + 2000), fastq, printable)
        for mode, doc, rep, bw in RUNS:
            o = subprocess.run([ORC_RUN, prefix, reads, mode, str(doc), str(rep), str(bw), "n", prefix + ".rawtext"], capture_output=True)
            assert o.returncode == 0, o.stderr.decode()
            exp = os.path.join(d, "expected_" + mode)
            os.makedirs(exp)
            for e in outputs(mode, doc, rep):
                assert os.path.getsize(reads + e) > 0, e
                shutil.move(reads + e, os.path.join(exp, "reads.fa" + e))
        size = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(d) for f in fs)
        print(name, "r =", raw.r, "n =", raw.n, "reads =", offs.size - 1, "bytes =", size)


if __name__ == "__main__":
    main()
