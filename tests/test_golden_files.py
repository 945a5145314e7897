"""Committed FILE-level fixtures (tests/golden/files/, made by tests/golden/make_golden_files.py): index files + a reads
file + the output files of `spumoni run`'s harness for them.  CPU: the oracle harness
(oracle/orc_run) still writes exactly these bytes.  The HIP-backed `spumoni run` is held against orc_run file by file
in tests/test_gpu_cli.py (-m gpu), on inputs of the same shapes."""
import filecmp
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = os.path.join(ROOT, "tests", "golden", "files")
ORC_RUN = os.path.join(ROOT, "oracle", "orc_run")
CASES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(FILES, "*")) if os.path.isdir(p))
RUNS = [("P", 1, 1, 50), ("M", 1, 1, 60)]  # as in make_golden_files.py


def test_fixtures_exist():
    assert {"dna_multiline_fasta", "dna_fastq", "promoted_alphabet_fasta"} <= set(CASES)


@pytest.mark.parametrize("case", CASES)
def test_oracle_harness_reproduces_the_committed_files(oracle_mod, tmp_path, case):
    work = tmp_path / case
    shutil.copytree(os.path.join(FILES, case), work)
    prefix, reads = str(work / "ref.fa"), str(work / "reads.fa")
    for mode, doc, rep, bw in RUNS:
        o = subprocess.run([ORC_RUN, prefix, reads, mode, str(doc), str(rep), str(bw), "n", prefix + ".rawtext"], capture_output=True)
        assert o.returncode == 0, o.stderr.decode()
        want = sorted(os.listdir(work / ("expected_" + mode)))
        assert want == sorted(["reads.fa" + e for e in
                               ([".pseudo_lengths"] if mode == "P" else [".lengths", ".pointers"]) + [".doc_numbers", ".report"]])
        for name in want:
            assert filecmp.cmp(str(work / name), str(work / ("expected_" + mode) / name), shallow=False), (case, mode, name)


def test_fastq_tail_batch_is_dropped_like_the_reference_drops_it():
    """BatchLoader::loadBatch returns false when getline meets the end of the file inside a batch
    (/root/reference/src/batch_loader.cpp:49-51): the last, incomplete 1000-base batch of a FASTQ file
    that ends in a newline is never processed.  FASTA is immune (peek() sets eofbit first, :68)."""
    def records(path, mark):
        return sum(1 for line in open(path, "rb") if line.startswith(mark))

    fq = os.path.join(FILES, "dna_fastq")
    n_in = records(os.path.join(fq, "reads.fa"), b"@read_")
    n_out = records(os.path.join(fq, "expected_P", "reads.fa.pseudo_lengths"), b">")
    assert 0 < n_out < n_in
    assert records(os.path.join(fq, "expected_M", "reads.fa.lengths"), b">") == n_out
    fa = os.path.join(FILES, "dna_multiline_fasta")
    assert records(os.path.join(fa, "reads.fa"), b">read_") == records(os.path.join(fa, "expected_P", "reads.fa.pseudo_lengths"), b">")


def test_committed_files_under_asan_ubsan(tmp_path):
    """The same inputs through the ASan + UBSan build of the harness (multi-line FASTA, FASTQ with a dropped tail,
    bytes >= 128 in the reads): no report, the same bytes."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "all", "san"], stdout=subprocess.DEVNULL)
    orc_asan = os.path.join(ROOT, "oracle", "orc_run_asan")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    for case in CASES:
        work = tmp_path / case
        shutil.copytree(os.path.join(FILES, case), work)
        prefix, reads = str(work / "ref.fa"), str(work / "reads.fa")
        for mode, doc, rep, bw in RUNS:
            o = subprocess.run([orc_asan, prefix, reads, mode, str(doc), str(rep), str(bw), "n", prefix + ".rawtext"], capture_output=True, env=env)
            assert o.returncode == 0, o.stderr.decode()[-2000:]
            for bad in (b"AddressSanitizer", b"runtime error:", b"LeakSanitizer"):
                assert bad not in o.stderr, o.stderr.decode()[-3000:]
            for name in os.listdir(work / ("expected_" + mode)):
                assert filecmp.cmp(str(work / name), str(work / ("expected_" + mode) / name), shallow=False), (case, mode, name)
