"""Sanitizer builds (hygiene): the host sources above the C-ABI under ASan + UBSan and TSan, the oracle
harness under ASan + UBSan.

CPU (here): the host binary's CPU modes -- read batching, the serialised-index / sdsl stream readers -- and the oracle
harness run clean under ASan + UBSan on the inputs of the ordinary tests, with the same output as the ordinary build.
GPU (-m gpu): the whole `spumoni run` harness (parser, one queue, a worker thread per device, the ordered pwrite
writer: spumoni_amd/csrc/host/classify.cpp) with two workers under TSan and under ASan, files identical to the
ordinary binary's."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "spumoni_amd", "bin")
HOST, HOST_ASAN, HOST_TSAN = (os.path.join(BIN, n) for n in ("spumoni", "spumoni_asan", "spumoni_tsan"))
ORC, ORC_ASAN = os.path.join(ROOT, "oracle", "orc_run"), os.path.join(ROOT, "oracle", "orc_run_asan")
BAD = (b"AddressSanitizer", b"runtime error:", b"ThreadSanitizer", b"LeakSanitizer")


@pytest.fixture(scope="module")
def san_bins():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "spumoni_amd", "csrc"), "-j4"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "spumoni_amd", "csrc", "host"), "all", "san", "-j2"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "all", "san"], stdout=subprocess.DEVNULL)


def _clean(r, what):
    assert r.returncode == 0, (what, r.stderr[-2000:])
    for b in BAD:
        assert b not in r.stderr, (what, r.stderr[-3000:])


def _env(leaks=1):
    e = dict(os.environ)
    e["ASAN_OPTIONS"] = f"detect_leaks={leaks}:abort_on_error=0"
    e["UBSAN_OPTIONS"] = "print_stacktrace=1"
    return e


def test_host_cpu_modes_under_asan_ubsan(san_bins, tmp_path):
    from tests.test_host_reads import _cases

    rng = np.random.default_rng(3)
    for name, text in _cases(rng).items():
        p = tmp_path / name
        p.write_bytes(text.encode())
        a = subprocess.run([HOST_ASAN, "dump-reads", str(p)], capture_output=True, env=_env())
        _clean(a, name)
        b = subprocess.run([HOST, "dump-reads", str(p)], capture_output=True)
        assert a.stdout == b.stdout, name
    # the serialised-index reader on a file written by the test writer, and on damaged copies of it
    from tests import cases, sdsl_files

    raw, _ = cases.real_case(7, 3000, list(b"ACGT"), ndocs=2)
    path = str(tmp_path / "ix.thrbv.spumoni")
    sdsl_files.write_thrbv(path, raw.heads.numpy(), raw.lens.numpy(), raw.thr.numpy())
    a = subprocess.run([HOST_ASAN, "dump-index", path, "P"], capture_output=True, env=_env())
    _clean(a, "dump-index")
    assert a.stdout == subprocess.run([HOST, "dump-index", path, "P"], capture_output=True).stdout
    blob = open(path, "rb").read()
    for cut in (len(blob) // 3, len(blob) - 5, 17):
        open(path, "wb").write(blob[:cut])
        r = subprocess.run([HOST_ASAN, "dump-index", path, "P"], capture_output=True, env=_env())
        assert r.returncode != 0  # refused ...
        for b in BAD:
            assert b not in r.stderr, r.stderr[-2000:]  # ... cleanly


def test_oracle_harness_under_asan_ubsan(san_bins, tmp_path):
    """The checker itself: orc_run file to file on a small real index, PML and MS with documents, -O3 build against
    the ASan + UBSan build."""
    from tests import cases
    from tests.sdsl_files import write_doc_array, write_null_db

    raw, text = cases.real_case(11, 4000, list(b"ACGT") + [ord("N")], ndocs=3)
    prefix = str(tmp_path / "ref.fa")
    raw.write_raw_files(prefix)
    text.tofile(prefix + ".rawtext")
    write_doc_array(prefix + ".doc", raw.doc_start.numpy(), raw.doc_end.numpy())
    write_null_db(prefix + ".pmlnulldb", 4.0, [1, 2, 3, 4, 4, 4])
    write_null_db(prefix + ".msnulldb", 9.0, [5, 9, 9, 9])
    rng = np.random.default_rng(2)
    seqs, offs = cases.reads_mixed(rng, text, list(b"ACGT"), 60, 300, [ord("N")])
    outs = {}
    for tag, binary in (("plain", ORC), ("asan", ORC_ASAN)):
        d = tmp_path / tag
        d.mkdir()
        with open(d / "reads.fa", "w") as f:
            for q in range(offs.size - 1):
                s = seqs[offs[q]: offs[q + 1]].tobytes().decode("latin-1")
                if s:
                    f.write(f">r{q}\n{s}\n")
        for mode in "PM":
            r = subprocess.run([binary, prefix, str(d / "reads.fa"), mode, "1", "1", "150", "n", prefix + ".rawtext"],
                               capture_output=True, env=_env(leaks=0))  # (a run-to-exit tool: it frees nothing at exit)
            _clean(r, tag + mode)
        outs[tag] = {n: open(d / n, "rb").read() for n in sorted(os.listdir(d))}
    assert outs["plain"].keys() == outs["asan"].keys() and len(outs["plain"]) >= 6
    assert outs["plain"] == outs["asan"]


def _our_reports(err):
    """Sanitizer reports whose racing / faulting access is in the host sources: in each access stack (the frames up
    to the report's "Location" / "Thread ... created by" part) the first frame outside the sanitizer runtime must lie
    in spumoni_amd/csrc/host.  The HIP / HSA runtimes race among their own threads under TSan; those reports name our
    files only as the place their threads were created from."""
    ours = []
    for blk in err.split("=================="):
        if "Sanitizer" not in blk:
            continue
        head = blk
        for mark in ("\n  Location is", "\n  Thread T", "\n  Mutex M"):
            head = head.split(mark)[0]
        stacks, cur = [], []
        for ln in head.splitlines():
            t = ln.strip()
            if t.startswith("#"):
                cur.append(t)
            elif cur:
                stacks.append(cur)
                cur = []
        if cur:
            stacks.append(cur)
        for st in stacks:
            first = next((f for f in st if "libtsan" not in f and "libasan" not in f and "sanitizer" not in f), "")
            if "csrc/host" in first:
                ours.append(blk)
                break
    return ours


@pytest.mark.gpu
@pytest.mark.timeout(1200)
@pytest.mark.parametrize("which", ["tsan", "asan"])
def test_run_harness_under_sanitizers(san_bins, tmp_path, which):
    """`spumoni run` with two workers on one device (SPUMONI_GPUS=0,0: two index replicas, one queue, the ordered
    writer) built with -fsanitize=thread / address: no report from the host sources, files identical to the ordinary
    binary's.  (Reports that name only the HIP runtime's own threads are not ours to fix: they are listed in the
    failure message but only frames of spumoni_amd/csrc/host fail the test.)"""
    from tests.test_gpu_cli import _setup, _write_fasta

    ref, prefix, seqs, offs, rng = _setup(tmp_path, 93, list(b"ACGT"), nreads=400)
    binary = HOST_TSAN if which == "tsan" else HOST_ASAN
    files = {}
    for tag, exe in (("plain", HOST), (which, binary)):
        d = tmp_path / tag
        shutil.rmtree(d, ignore_errors=True)
        d.mkdir()
        _write_fasta(d / "reads.fa", seqs, offs, np.random.default_rng(5))
        env = dict(os.environ, SPUMONI_GPUS="0,0", SPUMONI_CACHE="off", SPUMONI_TEXT=prefix + ".rawtext")
        env["TSAN_OPTIONS"] = "report_signal_unsafe=0:history_size=4:exitcode=0"
        env["ASAN_OPTIONS"] = "detect_leaks=0:protect_shadow_gap=0"
        for mode, flags in (("-P", ["-c", "-d"]), ("-M", ["-c", "-d"])):
            # (gcc 11's TSan runtime does not know this kernel's randomised mappings: run it with ASLR off)
            pre = ["setarch", "x86_64", "-R"] if (exe == HOST_TSAN and shutil.which("setarch")) else []
            r = subprocess.run(pre + [exe, "run", "-r", ref, "-p", str(d / "reads.fa"), "-n", mode] + flags, capture_output=True, env=env)
            if exe == HOST_TSAN and b"unexpected memory mapping" in r.stderr:
                pytest.skip("this TSan runtime cannot start on this kernel (unexpected memory mapping), with or without ASLR")
            assert r.returncode == 0, r.stderr.decode()[-3000:]
            ours = _our_reports(r.stderr.decode(errors="replace"))
            assert not ours, "\n".join(ours)[:6000]
        files[tag] = {n: open(d / n, "rb").read() for n in sorted(os.listdir(d)) if n != "reads.fa"}
    assert files["plain"].keys() == files[which].keys() and len(files["plain"]) >= 5
    assert files["plain"] == files[which]


def test_serialised_index_reader_survives_damaged_files(san_bins, tmp_path):
    """tools/index_reader_fuzz.py, 160 mutations of a valid <ref>.thrbv.spumoni (truncations, flipped bytes, wild size
    words) through the ASan + UBSan build: decoded or refused with a message, never a report, an abort or a hang.  (Its
    first run found size fields that sized 2^60-byte allocations before they were held against the file's length.)"""
    import sys

    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "index_reader_fuzz.py"), "160"], capture_output=True, text=True,
                       env=dict(os.environ, INDEX_FUZZ_DIR=str(tmp_path / "fuzz")), timeout=900)
    assert r.returncode == 0 and "bad 0" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
