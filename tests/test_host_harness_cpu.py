"""CPU tier: the C++ host above the C-ABI -- `spumoni run`'s harness (spumoni_amd/csrc/host: parser, one queue, a worker
thread per device, the ordered writer, the report thread), the pml_t / ms_t mirror -- on a machine without a GPU.

The host binary is the ordinary one; its libspumoni_gpu.so is, for these tests only, tests/fake_device: the same C-ABI
answered by the CPU oracle (test infrastructure: built into a temporary directory, found through LD_LIBRARY_PATH; the
product library still fails loudly without a device, tests/test_abi.py).  What is under test is the HOST code: batch
segmentation, FASTA / FASTQ parsing, super-batches through the queue and several workers, headers spliced into the text
the boundary returns, ordered writes, reports, fatal errors in read order, digestion options -- every output file
byte-identical to the oracle harness (oracle/orc_run), also with many small super-batches on three workers, and under
AddressSanitizer and ThreadSanitizer.  The HIP path itself is held against the oracle on the GPU (tests/test_gpu_*.py),
where the same CLI tests run against the real library.
"""
import filecmp
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "spumoni_amd", "bin")
FILES = os.path.join(ROOT, "tests", "golden", "files")


@pytest.fixture
def on_fake_device(fake_device, monkeypatch):
    monkeypatch.setenv("LD_LIBRARY_PATH", fake_device + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    monkeypatch.setenv("SPUMONI_CACHE", "off")
    monkeypatch.delenv("SPUMONI_GPUS", raising=False)
    return fake_device


def test_fake_device_exports_the_whole_boundary(fake_device):
    from spumoni_amd import capi

    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(fake_device, "libspumoni_gpu.so")], capture_output=True, text=True).stdout
    have = {ln.split()[-1] for ln in out.splitlines() if " T spx_" in ln}
    assert have == set(capi.EXPORTS)


# ---- the CLI tests of tests/test_gpu_cli.py, host side, against the fake device --------------------------------------------
def _cli():
    from tests import test_gpu_cli

    return test_gpu_cli


def test_cli_pml_and_ms_with_reports_and_documents(on_fake_device, tmp_path):
    T = _cli()
    T.test_cli_pml_report_doc(None, tmp_path)
    T.test_cli_ms_report_doc(None, tmp_path)


def test_cli_fastq_validation_and_serialised_index(on_fake_device, tmp_path):
    T = _cli()
    (tmp_path / "fq").mkdir()
    (tmp_path / "ser").mkdir()
    T.test_cli_fastq_content_in_fa_named_file(None, tmp_path / "fq")
    T.test_cli_validation_messages(None, tmp_path)
    T.test_cli_reads_the_serialised_index(None, tmp_path / "ser")


def test_cli_fatal_errors_come_in_read_order(on_fake_device, tmp_path, oracle_mod):
    T = _cli()
    (tmp_path / "a").mkdir()
    (tmp_path / "b").mkdir()
    T.test_cli_empty_read_is_fatal_after_earlier_reads_were_written(None, tmp_path / "a")
    T.test_cli_read_empty_after_digestion_is_fatal_in_order(None, tmp_path / "b", oracle_mod)


@pytest.mark.parametrize("digest,kw", [("m", ()), ("a", (2, 2))])
def test_cli_with_minimizer_digestion(on_fake_device, tmp_path, oracle_mod, digest, kw):
    _cli().test_cli_with_minimizer_digestion(None, tmp_path, oracle_mod, digest, kw)


def test_pml_t_ms_t_mirror(on_fake_device, tmp_path, oracle_mod):
    _cli().test_pml_t_ms_t_mirror_per_read_calls(None, tmp_path, oracle_mod)


@pytest.mark.parametrize("host_format", [False, True])
def test_many_small_super_batches_on_three_workers(on_fake_device, tmp_path, monkeypatch, host_format):
    """SPUMONI_SUPER_BATCH=3000 characters and SPUMONI_GPUS=0,0,0: some twenty super-batches dealt to three workers
    (three index replicas), results re-sequenced by the ordered writer and the report thread one batch behind -- the
    same bytes as the oracle harness, with the text from the boundary and (SPUMONI_HOST_FORMAT=1) formatted on the host."""
    T = _cli()
    monkeypatch.setenv("SPUMONI_SUPER_BATCH", "3000")
    monkeypatch.setenv("SPUMONI_GPUS", "0,0,0")
    if host_format:
        monkeypatch.setenv("SPUMONI_HOST_FORMAT", "1")
    ref, prefix, seqs, offs, rng = T._setup(tmp_path, 47, list(b"ACGT"), nreads=400)
    r = T._run_both(tmp_path, ref, prefix, "reads.fa", seqs, offs, rng, ["-c", "-d", "-w", "50"], "-P")
    batches = [int(ln.split("(")[1].split()[0]) for ln in r.stderr.decode().splitlines() if "super-batches" in ln]
    assert len(batches) == 3 and sum(batches) >= 10 and min(batches) >= 1, batches
    T._run_both(tmp_path, ref, prefix, "reads.fa", seqs, offs, rng, ["-c", "-d", "-w", "60"], "-M")
    T._run_both(tmp_path, ref, prefix, "reads.fa", seqs, offs, rng, [], "-P", fastq=True)


def test_default_devices_every_visible_device_twice(on_fake_device, tmp_path, monkeypatch):
    """Without SPUMONI_GPUS the CLI uses what it sees (the reference's OpenMP region over all threads,
    compute_ms_pml.cpp:890-1024): one visible device -> three workers on it; several -> every device, two workers each
    (FAKE_SPX_DEVICES=4: eight workers on devices 0..3, four copies of the index).  Same bytes as the oracle harness."""
    T = _cli()
    monkeypatch.setenv("SPUMONI_SUPER_BATCH", "3000")
    ref, prefix, seqs, offs, rng = T._setup(tmp_path, 49, list(b"ACGT"), nreads=600)
    r = T._run_both(tmp_path, ref, prefix, "reads.fa", seqs, offs, rng, ["-c", "-w", "50"], "-P")
    assert len([ln for ln in r.stderr.decode().splitlines() if "super-batches" in ln]) == 3
    monkeypatch.setenv("FAKE_SPX_DEVICES", "4")
    r = T._run_both(tmp_path, ref, prefix, "reads.fa", seqs, offs, rng, ["-c", "-w", "50"], "-P")
    batches = [int(ln.split("(")[1].split()[0]) for ln in r.stderr.decode().splitlines() if "super-batches" in ln]
    assert len(batches) == 8 and sum(batches) >= 16, batches
    monkeypatch.setenv("SPUMONI_GPUS", "2")  # (the override still rules)
    r = T._run_both(tmp_path, ref, prefix, "reads.fa", seqs, offs, rng, ["-c", "-w", "50"], "-P")
    assert len([ln for ln in r.stderr.decode().splitlines() if "super-batches" in ln]) == 1


def _limited(limit_bytes):
    """preexec_fn: no file may grow past limit_bytes; the kernel's SIGXFSZ is ignored, so the calls fail with EFBIG instead"""
    def fn():
        import resource
        import signal

        signal.signal(signal.SIGXFSZ, signal.SIG_IGN)
        resource.setrlimit(resource.RLIMIT_FSIZE, (limit_bytes, limit_bytes))
    return fn


@pytest.mark.parametrize("prep", ["populate", "falloc"])
def test_prepared_tail_that_cannot_be_had_falls_back_to_plain_writes(on_fake_device, tmp_path, monkeypatch, prep):
    """the files' tails are prepared from an ESTIMATE (here four times the truth, SPUMONI_MAP_FACTOR=4); where the
    file system cannot give that much (a full disk; here RLIMIT_FSIZE at twice the real size of the largest file) the preparation
    fails quietly and the run writes the ordinary way -- same bytes as the oracle harness (compute_ms_pml.cpp:1001-1021), nothing
    left behind.  Both ways of preparing: SPUMONI_PREP=populate (ftruncate + MADV_POPULATE_WRITE) and falloc (fallocate: the default)."""
    T = _cli()
    monkeypatch.setenv("SPUMONI_SUPER_BATCH", "3000")
    monkeypatch.setenv("SPUMONI_GPUS", "0,0,0")
    ref, prefix, seqs, offs, rng = T._setup(tmp_path, 50, list(b"ACGT"), nreads=400)
    r = T._run_both(tmp_path, ref, prefix, "reads.fa", seqs, offs, rng, ["-c", "-d", "-w", "50"], "-P")
    gpu = tmp_path / "gpu"
    real = max(os.path.getsize(gpu / f) for f in os.listdir(gpu))
    want = {f: open(gpu / f, "rb").read() for f in os.listdir(gpu) if not f.endswith(".fa")}
    for f in want:
        os.remove(gpu / f)
    env = dict(os.environ, SPUMONI_MAP_MIN="1", SPUMONI_MAP_FACTOR="4", SPUMONI_PREP=prep)
    cmd = [T.HOST_BIN, "run", "-r", ref, "-p", str(gpu / "reads.fa"), "-n", "-P", "-c", "-d", "-w", "50"]
    r = subprocess.run(cmd, capture_output=True, env=env, preexec_fn=_limited(2 * real))
    assert r.returncode == 0, r.stderr.decode()
    err = r.stderr.decode()
    assert "its tail was prepared as memory" not in [ln for ln in err.splitlines() if "writer lengths" in ln][0], err
    got = {f: open(gpu / f, "rb").read() for f in os.listdir(gpu) if not f.endswith(".fa")}
    assert got == want  # (nothing else in the directory either: no *.partial.* left)
    # ... and with room for the estimate the tail is memory again
    r = subprocess.run(cmd, capture_output=True, env=env, preexec_fn=_limited(64 * real))
    assert r.returncode == 0 and "its tail was prepared as memory" in r.stderr.decode()
    assert {f: open(gpu / f, "rb").read() for f in os.listdir(gpu) if not f.endswith(".fa")} == want


def test_leftovers_of_killed_and_interrupted_runs(on_fake_device, tmp_path, monkeypatch):
    """A run that is killed outright leaves its prepared `<output>.partial.<pid>` behind (nothing of it could clean up): the next run
    over the same pattern file removes the files of processes that are gone -- and only those.  A run that is interrupted (SIGTERM)
    removes its own on the way out."""
    import signal
    import time

    T = _cli()
    monkeypatch.setenv("SPUMONI_SUPER_BATCH", "3000")
    monkeypatch.setenv("SPUMONI_GPUS", "0,0,0")
    monkeypatch.setenv("SPUMONI_MAP_MIN", "1")
    ref, prefix, seqs, offs, rng = T._setup(tmp_path, 51, list(b"ACGT"), nreads=400)
    gpu = tmp_path / "gpu"
    dead = subprocess.Popen(["true"])
    dead.wait()
    stale = [gpu / f"reads.fa.pseudo_lengths.partial.{dead.pid}", gpu / f"reads.fa.report.old.{dead.pid}"]
    alive = gpu / f"reads.fa.pseudo_lengths.partial.{os.getpid()}"  # (this test's own process: alive)

    def plant():
        gpu.mkdir(exist_ok=True)
        for f in stale + [alive]:
            f.write_bytes(b"x" * 5000)

    # (_run_both recreates the directory: plant the files through a wrapper of the fasta writer it calls)
    orig = T._write_fasta

    def write_and_plant(path, *a, **k):
        orig(path, *a, **k)
        if str(path).startswith(str(gpu)):
            plant()

    monkeypatch.setattr(T, "_write_fasta", write_and_plant)
    T._run_both(tmp_path, ref, prefix, "reads.fa", seqs, offs, rng, ["-c", "-w", "50"], "-P")
    assert not any(f.exists() for f in stale) and alive.exists()
    alive.unlink()
    # interrupted: SPUMONI_TEST_STALL_MS holds the run after the files were prepared
    env = dict(os.environ, SPUMONI_TEST_STALL_MS="3000")
    p = subprocess.Popen([T.HOST_BIN, "run", "-r", ref, "-p", str(gpu / "reads.fa"), "-n", "-P", "-c"], env=env,
                         stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    mine = [gpu / f"reads.fa.pseudo_lengths.partial.{p.pid}", gpu / f"reads.fa.report.partial.{p.pid}"]
    t0 = time.time()
    while not mine[0].exists() and time.time() - t0 < 20:
        time.sleep(0.02)
    assert mine[0].exists(), "the run did not prepare its files"
    p.send_signal(signal.SIGTERM)
    assert p.wait(timeout=20) == 128 + signal.SIGTERM
    assert not any(f.exists() for f in mine)


@pytest.mark.parametrize("regime", [{"SPUMONI_MAP_MIN": "1"}, {"SPUMONI_MAP_MIN": "1", "SPUMONI_MAP_FACTOR": "0.3"},
                                    {"SPUMONI_MAP_MIN": "1", "SPUMONI_MAP_FACTOR": "0.02"},
                                    {"SPUMONI_MAP_MIN": "1", "SPUMONI_MAP_OUTPUT": "nopin", "SPUMONI_MAP_FACTOR": "0.6"},
                                    {"SPUMONI_MAP_MIN": "1", "SPUMONI_TRIM_MIN": "0", "SPUMONI_MAP_FACTOR": "4"},
                                    {"SPUMONI_MAP_MIN": "1", "SPUMONI_TRIM_MIN": "0", "SPUMONI_PIN_SHARE": "0.2"},
                                    {"SPUMONI_MAP_MIN": "1", "SPUMONI_PIN_SHARE": "1"},
                                    {"SPUMONI_MAP_OUTPUT": "0"}])
def test_output_files_tails_as_memory(on_fake_device, tmp_path, monkeypatch, oracle_mod, regime):
    """Round 5: the output files' tails are prepared as memory while the index loads (allocated, mapped, registered with the
    device) and a super-batch's text lands in the file's pages at its place in input order -- no write().  The same bytes as
    the oracle harness when every tail is mapped (SPUMONI_MAP_MIN=1: also for these tiny files), when the estimate is short
    and the run crosses into plain writes after a few super-batches (SPUMONI_MAP_FACTOR), when the mapping is not
    registered and the pool copies the text in (nopin), when the estimate is four times too long and the excess is cut off
    beside the run (EarlyTrim; SPUMONI_TRIM_MIN=0: for tiny files too), when only a fifth of the tail is registered with the
    device (SPUMONI_PIN_SHARE) or all of it, and with the mechanism off; a fatal read cuts the files where the reference
    stops although later super-batches are already in the mapping."""
    T = _cli()
    for k, v in regime.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("SPUMONI_SUPER_BATCH", "3000")
    monkeypatch.setenv("SPUMONI_GPUS", "0,0,0")
    ref, prefix, seqs, offs, rng = T._setup(tmp_path, 48, list(b"ACGT"), nreads=400)
    r = T._run_both(tmp_path, ref, prefix, "reads.fa", seqs, offs, rng, ["-c", "-d", "-w", "50"], "-P")
    err = r.stderr.decode()
    if regime.get("SPUMONI_MAP_OUTPUT") == "0":
        assert "its tail was prepared as memory" not in err
    else:
        assert "its tail was prepared as memory" in err, err[-1500:]
    if regime.get("SPUMONI_MAP_FACTOR") == "4":
        cut = [ln for ln in err.splitlines() if "cut off beside the run:" in ln]
        assert cut and float(cut[0].split("cut off beside the run:")[1].split()[0]) > 0, err[-1500:]
    T._run_both(tmp_path, ref, prefix, "reads.fa", seqs, offs, rng, ["-c", "-d", "-w", "60"], "-M")
    (tmp_path / "a").mkdir()
    (tmp_path / "b").mkdir()
    T.test_cli_empty_read_is_fatal_after_earlier_reads_were_written(None, tmp_path / "a")
    T.test_cli_read_empty_after_digestion_is_fatal_in_order(None, tmp_path / "b", oracle_mod)
    assert not [f for f in os.listdir(tmp_path) if ".partial." in f]


@pytest.mark.parametrize("case", sorted(os.listdir(FILES)) if os.path.isdir(FILES) else [])
def test_cli_reproduces_the_committed_files(on_fake_device, tmp_path, monkeypatch, case):
    """tests/golden/files through the host binary: the harness's output files are the committed ones."""
    work = tmp_path / case
    shutil.copytree(os.path.join(FILES, case), work)
    monkeypatch.setenv("SPUMONI_TEXT", str(work / "ref.fa.rawtext"))
    monkeypatch.setenv("SPUMONI_SUPER_BATCH", "1500")
    for mode, bw in (("P", "50"), ("M", "60")):  # (as tests/golden/make_golden_files.py ran the oracle harness)
        r = subprocess.run([os.path.join(BIN, "spumoni"), "run", "-r", str(work / "ref"), "-p", str(work / "reads.fa"), "-n", "-" + mode, "-c", "-d",
                            "-w", bw], capture_output=True)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        for name in sorted(os.listdir(work / ("expected_" + mode))):
            assert filecmp.cmp(str(work / name), str(work / ("expected_" + mode) / name), shallow=False), (case, mode, name)


# ---- the harness under sanitizers, no GPU runtime in the process ------------------------------------------------------------
@pytest.mark.timeout(900)
@pytest.mark.parametrize("which", ["asan", "tsan"])
def test_harness_under_sanitizers_on_the_fake_device(on_fake_device, tmp_path, which):
    """Three workers, many small super-batches, PML and MS with documents and reports, under -fsanitize=address,undefined
    and -fsanitize=thread: no report at all (no HIP / HSA runtime threads in the process here, so every report would be
    the host's), files identical to the ordinary binary's."""
    T = _cli()
    ref, prefix, seqs, offs, rng = T._setup(tmp_path, 94, list(b"ACGT"), nreads=300)
    exe = os.path.join(BIN, "spumoni_" + which)
    files = {}
    for tag, binary in (("plain", os.path.join(BIN, "spumoni")), (which, exe)):
        d = tmp_path / tag
        d.mkdir()
        T._write_fasta(d / "reads.fa", seqs, offs, np.random.default_rng(5))
        env = dict(os.environ, SPUMONI_GPUS="0,0,0", SPUMONI_SUPER_BATCH="4000", SPUMONI_TEXT=prefix + ".rawtext")
        env["TSAN_OPTIONS"] = "report_signal_unsafe=0:history_size=4:exitcode=66"
        # (no leak check: the slots' page-locked buffers outlive classify_reads on purpose, classify.cpp: the process ends)
        env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=0"
        env["UBSAN_OPTIONS"] = "print_stacktrace=1"
        for mode in ("-P", "-M"):
            pre = ["setarch", "x86_64", "-R"] if (binary.endswith("_tsan") and shutil.which("setarch")) else []
            r = subprocess.run(pre + [binary, "run", "-r", ref, "-p", str(d / "reads.fa"), "-n", mode, "-c", "-d"], capture_output=True, env=env)
            if binary.endswith("_tsan") and b"unexpected memory mapping" in r.stderr:
                pytest.skip("this TSan runtime cannot start on this kernel (unexpected memory mapping), with or without ASLR")
            assert r.returncode == 0, r.stderr.decode(errors="replace")[-4000:]
            for bad in (b"AddressSanitizer", b"runtime error:", b"ThreadSanitizer", b"LeakSanitizer"):
                assert bad not in r.stderr, r.stderr.decode(errors="replace")[-6000:]
        files[tag] = {n: open(d / n, "rb").read() for n in sorted(os.listdir(d)) if n != "reads.fa"}
    assert files["plain"].keys() == files[which].keys() and len(files["plain"]) >= 5
    assert files["plain"] == files[which]


def test_long_read_and_report_only_on_both_formatting_paths(on_fake_device, tmp_path):
    """A read of 70 000 characters (no 16-bit values for its super-batch) between short ones, PML and MS with documents and
    reports, text from the boundary and formatted on the host, under ASan: the oracle harness's bytes.  SPUMONI_REPORT_ONLY=1
    empties <pattern>.pseudo_lengths only (PML; with -M the lengths are what the report is made from: no effect) -- on both
    paths alike (the host-formatting path used to leave .lengths empty with -M)."""
    T = _cli()
    ref, prefix, seqs, offs, rng = T._setup(tmp_path, 96, list(b"ACGT"), n=90000, nreads=30)
    text = np.fromfile(prefix + ".rawtext", dtype=np.uint8)
    reads = str(tmp_path / "reads.fa")
    with open(reads, "wb") as f:
        f.write(b">short_1\n" + text[100:400].tobytes() + b"\n")
        f.write(b">long one\n" + text[2000:72000].tobytes() + b"\n")
        f.write(b">short_2 x\n" + text[500:900].tobytes().lower() + b"\n")
    exe = os.path.join(BIN, "spumoni_asan")
    for mode, exts in (("P", (".pseudo_lengths", ".doc_numbers", ".report")), ("M", (".lengths", ".pointers", ".doc_numbers", ".report"))):
        o = subprocess.run([T.ORC_RUN, prefix, reads, mode, "1", "1", "150", "n", prefix + ".rawtext"], capture_output=True)
        assert o.returncode == 0, o.stderr.decode()
        for e in exts:
            shutil.move(reads + e, str(tmp_path / ("want" + e)))
        for host_format in (False, True):
            for report_only in (False, True):
                env = dict(os.environ, SPUMONI_TEXT=prefix + ".rawtext", ASAN_OPTIONS="detect_leaks=0")
                if host_format:
                    env["SPUMONI_HOST_FORMAT"] = "1"
                if report_only:
                    env["SPUMONI_REPORT_ONLY"] = "1"
                r = subprocess.run([exe, "run", "-r", ref, "-p", reads, "-n", "-" + mode, "-c", "-d"], capture_output=True, env=env)
                assert r.returncode == 0 and b"Sanitizer" not in r.stderr and b"runtime error" not in r.stderr, r.stderr.decode(errors="replace")[-3000:]
                for e in exts:
                    if report_only and e == ".pseudo_lengths":
                        assert os.path.getsize(reads + e) == 0
                    else:
                        assert filecmp.cmp(reads + e, str(tmp_path / ("want" + e)), shallow=False), (mode, host_format, report_only, e)


@pytest.mark.timeout(900)
def test_cli_differential_fuzz_against_the_oracle_harness(on_fake_device, tmp_path):
    """tools/cli_fuzz_cpu.py, forty seeds: random FASTA / FASTQ files (headers without an id, empty and multi-line reads,
    CR LF, blank lines, no final newline, ...) through the ASan build of `spumoni run` on the fake device and through the
    oracle harness -- same exit status, same error message, same bytes in every file, also after a fatal error.
    (1 400 further seeds: profiles/r03_cli_fuzz_cpu.txt)"""
    env = dict(os.environ, FAKE_DEVICE_DIR=on_fake_device, CLI_FUZZ_DIR=str(tmp_path / "fuzz"))
    r = subprocess.run([os.sys.executable, os.path.join(ROOT, "tools", "cli_fuzz_cpu.py"), "40", "0"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "bad 0" in r.stdout


@pytest.mark.timeout(900)
@pytest.mark.parametrize("regime", [{"SPUMONI_MAP_MIN": "1"}, {"SPUMONI_MAP_MIN": "1", "SPUMONI_MAP_FACTOR": "0.3"},
                                    {"SPUMONI_MAP_MIN": "1", "SPUMONI_MAP_OUTPUT": "nopin", "SPUMONI_MAP_FACTOR": "0.5"},
                                    {"SPUMONI_MAP_MIN": "1", "SPUMONI_TRIM_MIN": "0", "SPUMONI_MAP_FACTOR": "2"}])
def test_cli_differential_fuzz_with_the_output_tails_as_memory(on_fake_device, tmp_path, regime):
    """The same fuzz, thirty other seeds, with the round-5 drain in each of its regimes (the files' tails mapped + registered,
    an estimate that runs out mid-run, mapped without registration, the excess cut off beside the run): fatal reads must still cut every file where the
    reference stops.  (6 000 further seeds, ASan and TSan builds: DESIGN.md 6.)"""
    env = dict(os.environ, FAKE_DEVICE_DIR=on_fake_device, CLI_FUZZ_DIR=str(tmp_path / "fuzz"), **regime)
    r = subprocess.run([os.sys.executable, os.path.join(ROOT, "tools", "cli_fuzz_cpu.py"), "30", "500"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "bad 0" in r.stdout


def test_general_text_mode(on_fake_device, tmp_path):
    """`run -g -n` (classify_general_reads_pml / _ms, compute_ms_pml.cpp:1219-1297) through the ASan build against the oracle
    harness: tests/test_gpu_cli.py::_general_text_case (bytes >= 128, a NUL, lower case, empty reads, trailing text,
    several super-batches, PML and MS)."""
    T = _cli()
    ref = T._general_text_case(tmp_path, os.path.join(BIN, "spumoni_asan"))
    # the option checks of include/spumoni_main.hpp:300-310
    for extra, msg in ((["-a"], b"minimizer digestion must be turned off"), (["-n", "-c"], b"classification is not available"),
                       (["-n", "-t", "2"], b"multi-threading is not available")):
        r = subprocess.run([os.path.join(BIN, "spumoni"), "run", "-r", ref, "-p", str(tmp_path / "cli" / "pattern.txt"), "-g", "-P"] + extra, capture_output=True)
        assert r.returncode == 1 and msg in r.stderr, r.stderr.decode()[-500:]


def test_the_gpu_tiers_new_multi_batch_test_host_side(on_fake_device, tmp_path, monkeypatch):
    """tests/test_gpu_cli.py::test_cli_many_small_super_batches_on_the_device (not yet run on a GPU: xfail-marked there)
    passes host side."""
    fn = _cli().test_cli_many_small_super_batches_on_the_device
    getattr(fn, "__wrapped__", fn)(None, tmp_path, monkeypatch)


def test_builder_files_run_end_to_end(on_fake_device, tmp_path, monkeypatch):
    """FASTA files -> spumoni_amd.build_index (suffix sorting with torch on the CPU here; the null reads' statistics through
    the binding, i.e. the fake device) -> `spumoni run` -> the oracle harness's files; the null reads are the reference's
    draws and both null databases re-derive from the oracle harness's output
    (tests/test_gpu_cli.py::test_end_to_end_from_fasta_with_our_builder, host side)."""
    monkeypatch.setenv("SPUMONI_GPU_LIB", os.path.join(on_fake_device, "libspumoni_gpu.so"))
    _cli().test_end_to_end_from_fasta_with_our_builder(None, tmp_path)


def test_minimizer_builder_files_run_end_to_end(on_fake_device, tmp_path, monkeypatch):
    """FASTA -> build_index -m (digestion through the binding) -> run -m: positives FOUND, nulls not
    (tests/test_gpu_cli.py::test_end_to_end_minimizer_index_from_fasta, host side)."""
    monkeypatch.setenv("SPUMONI_GPU_LIB", os.path.join(on_fake_device, "libspumoni_gpu.so"))
    _cli().test_end_to_end_minimizer_index_from_fasta(None, tmp_path)
