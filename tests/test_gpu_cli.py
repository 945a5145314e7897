"""-m gpu: the HIP-backed `spumoni run` binary against the oracle harness (oracle/orc_run):
every output file must be byte-identical."""
import filecmp
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch

from spumoni_amd import synth
from tests import cases

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_BIN = os.path.join(ROOT, "spumoni_amd", "bin", "spumoni")
ORC_RUN = os.path.join(ROOT, "oracle", "orc_run")


def _write_fasta(path, seqs, offs, rng, fastq=False):
    with open(path, "w") as f:
        for q in range(offs.size - 1):
            s = seqs[offs[q] : offs[q + 1]].tobytes().decode("latin-1")
            if not s:
                continue
            if rng.random() < 0.3:
                s = s.lower()
            name = f"read_{q}" + (" some description" if q % 5 == 0 else "")
            if fastq:
                f.write(f"@{name}\n{s}\n+\n{'I' * len(s)}\n")
            else:
                cut = int(rng.integers(1, len(s) + 1))
                f.write(f">{name}\n{s[:cut]}\n" + (f"{s[cut:]}\n" if cut < len(s) else ""))


def _setup(tmp_path, seed, letters, n=6000, nreads=300):
    raw, text = cases.real_case(seed, n, letters, ndocs=4)
    ref = str(tmp_path / "ref")
    open(ref + ".fa", "w").write(">dummy\n")  # run's validate() insists on <ref>.fa existing
    prefix = ref + ".fa"
    raw.write_raw_files(prefix)
    text.tofile(prefix + ".rawtext")
    # side files: .doc, .pmlnulldb / .msnulldb (sdsl framing written by our host writers via a helper)
    from tests.sdsl_files import write_doc_array, write_null_db

    write_doc_array(prefix + ".doc", raw.doc_start.numpy(), raw.doc_end.numpy())
    write_null_db(prefix + ".pmlnulldb", 4.0, [1, 2, 3, 4, 4, 4, 4, 4])
    write_null_db(prefix + ".msnulldb", 9.0, [5, 9, 9, 9, 9, 9])
    rng = np.random.default_rng(seed)
    seqs, offs = cases.reads_mixed(rng, text, letters, nreads, 400, [ord("N")])
    return ref, prefix, seqs, offs, rng


def _run_both(tmp_path, ref, prefix, reads_name, seqs, offs, rng, flags, mode, fastq=False, digest="n", kw=(), give_text=True):
    a_dir, b_dir = tmp_path / "gpu", tmp_path / "orc"
    for d in (a_dir, b_dir):
        shutil.rmtree(d, ignore_errors=True)
        d.mkdir()
    _write_fasta(a_dir / reads_name, seqs, offs, np.random.default_rng(77), fastq)
    shutil.copy(a_dir / reads_name, b_dir / reads_name)
    env = dict(os.environ)
    env.pop("SPUMONI_TEXT", None)
    if give_text:  # otherwise `run -M` rebuilds the text from the MS index itself
        env["SPUMONI_TEXT"] = prefix + ".rawtext"
    cmd = [HOST_BIN, "run", "-r", ref, "-p", str(a_dir / reads_name), "-" + digest, mode] + flags
    orc_kw = []
    if kw:
        cmd += ["-K", str(kw[0]), "-W", str(kw[1])]
        orc_kw = ["--k", str(kw[0]), "--w", str(kw[1])]
    r = subprocess.run(cmd, capture_output=True, env=env)
    assert r.returncode == 0, r.stderr.decode()
    assert b"finished processing" in r.stderr
    doc = "1" if "-d" in flags else "0"
    rep = "1" if "-c" in flags else "0"
    bw = flags[flags.index("-w") + 1] if "-w" in flags else "150"
    o = subprocess.run([ORC_RUN, prefix, str(b_dir / reads_name), mode[1], doc, rep, bw, digest, prefix + ".rawtext"]
                       + orc_kw, capture_output=True)
    assert o.returncode == 0, o.stderr.decode()
    exts = [".pseudo_lengths"] if mode == "-P" else [".lengths", ".pointers"]
    if doc == "1":
        exts.append(".doc_numbers")
    if rep == "1":
        exts.append(".report")
    for e in exts:
        fa, fb = str(a_dir / reads_name) + e, str(b_dir / reads_name) + e
        assert os.path.getsize(fb) > 0
        assert filecmp.cmp(fa, fb, shallow=False), e
    return r


@pytest.fixture(scope="module")
def built(built_all):
    assert torch.cuda.is_available()
    assert os.path.exists(HOST_BIN) and os.path.exists(ORC_RUN)


def test_cli_pml_report_doc(built, tmp_path):
    ref, prefix, seqs, offs, rng = _setup(tmp_path, 41, list(b"ACGT"))
    r = _run_both(tmp_path, ref, prefix, "reads.fa", seqs, offs, rng, ["-c", "-d", "-w", "50"], "-P")
    assert b"loading the PML index" in r.stderr and b"no minimizer digestion" in r.stderr


def test_cli_ms_report_doc(built, tmp_path):
    ref, prefix, seqs, offs, rng = _setup(tmp_path, 42, list(b"ACGT"))
    _run_both(tmp_path, ref, prefix, "reads.fa", seqs, offs, rng, ["-c", "-d", "-w", "60"], "-M")


def test_cli_fastq_content_in_fa_named_file(built, tmp_path):
    # validate() insists on a .fa name even for FASTQ content (include/spumoni_main.hpp:288-290)
    ref, prefix, seqs, offs, rng = _setup(tmp_path, 43, list(b"ACGT"), nreads=120)
    _run_both(tmp_path, ref, prefix, "reads.fa", seqs, offs, rng, ["-c"], "-P", fastq=True)


def test_cli_validation_messages(built, tmp_path):
    r = subprocess.run([HOST_BIN, "run", "-r", str(tmp_path / "nope"), "-p", str(tmp_path / "x.fa"), "-P", "-n"],
                       capture_output=True)
    assert r.returncode == 1 and b"The following path is not valid" in r.stderr
    r = subprocess.run([HOST_BIN, "run", "-r", "a", "-p", "b"], capture_output=True)
    assert r.returncode == 1 and b"An output type with -M or -P must be specified" in r.stderr


def test_pml_t_ms_t_mirror_per_read_calls(built, tmp_path, oracle_mod):
    """host/spumoni_index.hpp: the reference's pml_t / ms_t interface, one call per read."""
    shim = HOST_BIN.replace("bin/spumoni", "bin/shim_check")
    raw, text = cases.real_case(44, 3000, list(b"ACGT"), ndocs=3)
    prefix = str(tmp_path / "idx")
    raw.write_raw_files(prefix)
    text.tofile(prefix + ".rawtext")
    from tests.sdsl_files import write_doc_array

    write_doc_array(prefix + ".doc", raw.doc_start.numpy(), raw.doc_end.numpy())
    rng = np.random.default_rng(4)
    seqs, offs = cases.reads_mixed(rng, text, list(b"ACGT"), 40, 80, [ord("N")])
    reads = [seqs[offs[q] : offs[q + 1]].tobytes().decode() for q in range(offs.size - 1)]
    reads = [r for r in reads if r]
    (tmp_path / "reads.txt").write_text("\n".join(reads) + "\n")
    orc = oracle_mod.OracleIndex.from_raw(raw)
    s2 = np.frombuffer("".join(reads).encode(), dtype=np.uint8)
    o2 = np.concatenate([[0], np.cumsum([len(r) for r in reads])])
    pml, pdocs = orc.pml(s2, o2, want_docs=True)
    ms = orc.ms(s2, o2, want_docs=True, text=text)
    for mode in ("P", "M"):
        out = subprocess.run([shim, prefix, str(tmp_path / "reads.txt"), mode, "1"], capture_output=True)
        assert out.returncode == 0, out.stderr.decode()
        lines = out.stdout.decode().splitlines()
        assert lines[0] == f"stats {raw.n} {raw.r}"
        got = {"L": [], "P": [], "D": []}
        for ln in lines[1:]:
            tag, *vals = ln.split()
            got[tag].extend(int(v) for v in vals)
        if mode == "P":
            assert got["L"] == pml.tolist() and got["D"] == pdocs.tolist()
        else:
            assert got["L"] == ms["lengths"].tolist() and got["P"] == ms["pointers"].tolist()
            assert got["D"] == ms["docs"].tolist()


def test_cli_two_device_slots_same_output(built, tmp_path):
    """SPUMONI_GPUS=0,0: two index replicas + two host threads + sharded super-batches must give
    byte-identical files (the multi-GPU host path, exercised on one GPU)."""
    ref, prefix, seqs, offs, rng = _setup(tmp_path, 45, list(b"ACGT"), nreads=400)
    os.environ["SPUMONI_GPUS"] = "0,0"
    try:
        _run_both(tmp_path, ref, prefix, "reads.fa", seqs, offs, rng, ["-c", "-d"], "-P")
    finally:
        del os.environ["SPUMONI_GPUS"]


def test_cli_reads_the_serialised_index(built, tmp_path):
    """Only <ref>.fa.thrbv.spumoni present (no raw run files): the CLI must load it through the
    (unverified-layout) serialised-index reader and still produce the oracle's files."""
    from tests.sdsl_files import write_thrbv

    ref, prefix, seqs, offs, rng = _setup(tmp_path, 46, list(b"ACGT"), nreads=150)
    raw_dir = tmp_path / "rawcopy"
    raw_dir.mkdir()
    import glob

    heads = np.fromfile(prefix + ".bwt.heads", dtype=np.uint8)
    five = lambda p: np.frombuffer(open(p, "rb").read(), dtype=np.uint8).reshape(-1, 5)  # noqa: E731
    to_u64 = lambda a: (a.astype(np.uint64) << (8 * np.arange(5, dtype=np.uint64))).sum(1)  # noqa: E731
    lens, thr = to_u64(five(prefix + ".bwt.len")), to_u64(five(prefix + ".thr_pos"))
    write_thrbv(prefix + ".thrbv.spumoni", np.maximum(heads, 1), lens, thr)
    # the oracle harness keeps using the raw files: move them aside for the CLI run
    for f in glob.glob(prefix + ".bwt.*") + [prefix + ".thr_pos", prefix + ".ssa", prefix + ".esa"]:
        shutil.copy(f, raw_dir / os.path.basename(f))
    a_dir = tmp_path / "gpu"
    shutil.rmtree(a_dir, ignore_errors=True)
    a_dir.mkdir()
    _write_fasta(a_dir / "reads.fa", seqs, offs, np.random.default_rng(77))
    for f in glob.glob(prefix + ".bwt.*") + [prefix + ".thr_pos"]:
        os.remove(f)
    r = subprocess.run([HOST_BIN, "run", "-r", ref, "-p", str(a_dir / "reads.fa"), "-n", "-P", "-c"], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    b_dir = tmp_path / "orc"
    shutil.rmtree(b_dir, ignore_errors=True)
    b_dir.mkdir()
    shutil.copy(a_dir / "reads.fa", b_dir / "reads.fa")
    for f in os.listdir(raw_dir):
        shutil.copy(raw_dir / f, os.path.dirname(prefix) + "/" + f)
    o = subprocess.run([ORC_RUN, prefix, str(b_dir / "reads.fa"), "P", "0", "1", "150", "n"], capture_output=True)
    assert o.returncode == 0, o.stderr.decode()
    for e in (".pseudo_lengths", ".report"):
        assert filecmp.cmp(str(a_dir / "reads.fa") + e, str(b_dir / "reads.fa") + e, shallow=False), e


def test_end_to_end_from_fasta_with_our_builder(built, tmp_path):
    """FASTA files -> spumoni_amd.build_index (tooling) -> `spumoni run` -> files identical to the
    oracle harness run on the same index files; positives classify FOUND."""
    rng = np.random.default_rng(8)
    g1 = synth.random_genome(30_000, seed=21)
    g2 = synth.mutate(g1, seed=22)
    for name, g in (("a.fa", g1), ("b.fa", g2)):
        with open(tmp_path / name, "w") as f:
            f.write(f">{name}\n")
            s = g.tobytes().decode()
            for i in range(0, len(s), 70):
                f.write(s[i : i + 70] + "\n")
    (tmp_path / "list.txt").write_text(f"{tmp_path / 'a.fa'} 1\n{tmp_path / 'b.fa'} 2\n")
    ref = str(tmp_path / "idx" / "pan")
    r = subprocess.run(["python", "-m", "spumoni_amd.build_index", "-l", str(tmp_path / "list.txt"), "-o", ref, "--doc"],
                       capture_output=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()
    prefix = ref + ".fa"
    text = np.fromfile(prefix + ".rawtext", dtype=np.uint8)
    seqs, offs = synth.sample_reads(text, 400, 180, seed=3)
    for mode, flags in (("-P", ["-c", "-d"]), ("-M", ["-c", "-d"])):
        _run_both(tmp_path, ref, prefix, "reads.fa", seqs, offs, rng, flags, mode)
    rep = open(tmp_path / "gpu" / "reads.fa.report").read().splitlines()[1:]
    found = sum("FOUND" in ln and "NOT_PRESENT" not in ln for ln in rep)
    assert 0.3 * len(rep) < found < 0.7 * len(rep)  # sampled reads FOUND, reversed (null) reads not
    _check_null_databases(tmp_path, prefix, [g1, g2])


def _values_file(path):
    out = []
    for ln in open(path):
        if not ln.startswith(">"):
            out.append(np.array(ln.split(), dtype=np.int64))
    return out


def _check_null_databases(tmp_path, prefix, genomes):
    """The builder's null reads are the ones src/refbuilder.cpp:83-127 draws (100 per sequence, glibc rand() after srand(0));
    the databases hold the oracle harness's statistics of those reads reversed (compute_ms_pml.cpp:1409-1506), and the
    KS threshold is mean + 3 sd of the reads' own windows against the database, MS first (:1549-1663,
    src/spumoni.cpp:650-694)."""
    import struct
    from spumoni_amd import build_index as B

    null_path = os.path.join(os.path.dirname(prefix), "spumoni_null_reads.fa")
    lines = open(null_path).read().split("\n")
    names, reads = lines[0:-1:2], lines[1::2]
    assert names == [f">read_{i}" for i in range(200)] and all(len(r) == 150 for r in reads)
    g = B.GlibcRand(0)
    for i, rd in enumerate(reads):
        gen = genomes[i // 100]
        at = g.rand() % (gen.size - 150)
        assert rd == gen[at: at + 150].tobytes().decode(), i
    d = tmp_path / "nullchk"
    d.mkdir()
    with open(d / "rev.fa", "w") as f:
        for i, rd in enumerate(reads):
            f.write(f">read_{i}\n{rd[::-1]}\n")
    per_read = {}
    for mode, ext in (("M", ".lengths"), ("P", ".pseudo_lengths")):
        o = subprocess.run([ORC_RUN, prefix, str(d / "rev.fa"), mode, "0", "0", "150", "n", prefix + ".rawtext"], capture_output=True)
        assert o.returncode == 0, o.stderr.decode()
        per_read[mode] = _values_file(str(d / "rev.fa") + ext)
    for mode, ext in (("M", ".msnulldb"), ("P", ".pmlnulldb")):  # the same rand() sequence runs on from the reads through MS into PML
        vals = np.concatenate(per_read[mode])
        blob = open(prefix + ext, "rb").read()
        num, ks, mean, pct = struct.unpack("<Qddd", blob[:32])
        bits, width = struct.unpack("<QB", blob[32:41])
        assert num == vals.size == 200 * 150 and bits == num * width and width == B._width(vals.tolist())
        assert mean == pytest.approx(vals.mean(), rel=1e-12) and pct == B.percentile_value(vals)
        words = np.frombuffer(blob[41:], dtype="<u8")
        stored = np.array([(int(words[(i * width) >> 6]) >> ((i * width) & 63) | (int(words[((i * width) >> 6) + 1]) << (64 - ((i * width) & 63)) if ((i * width) & 63) + width > 64 else 0)) & ((1 << width) - 1)
                           for i in range(num)], dtype=np.int64)
        assert np.array_equal(stored, vals & ((1 << width) - 1))
        assert ks == B.ks_threshold(per_read[mode], stored, 150, g) and 0 < ks < 1, ext


def test_cli_empty_read_is_fatal_after_earlier_reads_were_written(built, tmp_path):
    """Appendix C13: a read that is empty is fatal when the reference reaches it -- everything
    before it has been written by then.  Same files, same exit code, "\\n\\n" on stdout."""
    ref, prefix, seqs, offs, rng = _setup(tmp_path, 47, list(b"ACGT"), nreads=40)
    for d in ("gpu", "orc"):
        (tmp_path / d).mkdir(exist_ok=True)
        with open(tmp_path / d / "reads.fa", "w") as f:
            for q in range(12):
                s = seqs[offs[q] : offs[q + 1]].tobytes().decode()
                if s:
                    f.write(f">r{q}\n{s}\n")
            f.write(">bad_one\n>after\nACGTACGT\n")
    r = subprocess.run([HOST_BIN, "run", "-r", ref, "-p", str(tmp_path / "gpu" / "reads.fa"), "-n", "-P", "-c"],
                       capture_output=True)
    o = subprocess.run([ORC_RUN, prefix, str(tmp_path / "orc" / "reads.fa"), "P", "0", "1", "150", "n"], capture_output=True)
    assert r.returncode == 1 and o.returncode == 1
    assert b"bad_one was empty after digestion" in r.stderr and b"bad_one was empty after digestion" in o.stderr
    assert r.stdout.endswith(b"\n\n")
    for e in (".pseudo_lengths", ".report"):
        a, b = str(tmp_path / "gpu" / "reads.fa") + e, str(tmp_path / "orc" / "reads.fa") + e
        assert os.path.getsize(b) > 0 and filecmp.cmp(a, b, shallow=False), e


def _setup_digested(tmp_path, oracle_mod, kind, k, w, seed):
    """Raw index files over the digestion of a repetitive genome (prefix <ref>.bin for -m, <ref>.fa
    for -a, src/spumoni.cpp:744-747) + DNA reads, some with N runs."""
    from tests.sdsl_files import write_doc_array, write_null_db

    rng = np.random.default_rng(seed)
    genome = cases.repetitive_text(rng, 40000, list(b"ACGT"))
    dtext = oracle_mod.digest(kind, k, w, genome)
    raw = synth.index_from_text(torch.from_numpy(dtext.copy()), doc_lengths=[dtext.size // 3, dtext.size - dtext.size // 3])
    ref = str(tmp_path / "ref")
    prefix = ref + (".bin" if kind == 1 else ".fa")
    open(prefix, "w").write(">dummy\n")
    raw.write_raw_files(prefix)
    dtext.tofile(prefix + ".rawtext")
    write_doc_array(prefix + ".doc", raw.doc_start.numpy(), raw.doc_end.numpy())
    write_null_db(prefix + ".pmlnulldb", 4.0, [1, 2, 3, 4, 4, 4, 4, 4])
    write_null_db(prefix + ".msnulldb", 9.0, [5, 9, 9, 9, 9, 9])
    seqs, offs = cases.reads_mixed(rng, genome, list(b"ACGT"), 300, 500, [ord("N")])
    # every read must survive digestion here (the fatal case has its own test): drop short ones
    keep = [q for q in range(offs.size - 1) if len(oracle_mod.digest(kind, k, w, seqs[offs[q]:offs[q + 1]])) > 0]
    reads = [seqs[offs[q]:offs[q + 1]] for q in keep]
    offs2 = np.concatenate([[0], np.cumsum([r.size for r in reads])]).astype(np.int64)
    return ref, prefix, np.concatenate(reads), offs2, rng


@pytest.mark.parametrize("digest,kw", [("m", ()), ("a", ()), ("m", (3, 9)), ("a", (2, 2))])
def test_cli_with_minimizer_digestion(built, tmp_path, oracle_mod, digest, kw):
    """`run -m` / `run -a`: reads are digested (on the device) before the walk, outputs are per
    digested character (compute_ms_pml.cpp:919-938) -- byte-identical to the oracle harness."""
    kind = 1 if digest == "m" else 2
    k, w = kw if kw else (4, 11)
    ref, prefix, seqs, offs, rng = _setup_digested(tmp_path, oracle_mod, kind, k, w, seed=60 + kind)
    assert offs.size > 100
    for mode, flags in (("-P", ["-c", "-d", "-w", "50"]), ("-M", ["-c", "-d", "-w", "50"]), ("-P", [])):
        r = _run_both(tmp_path, ref, prefix, "reads.fa", seqs, offs, rng, flags, mode, digest=digest, kw=kw)
        assert (b"promoted minimizer alphabet" if digest == "m" else b"DNA minimizer alphabet") in r.stderr


def test_cli_walks_the_digested_reads_where_they_were_parked(built, tmp_path, oracle_mod, monkeypatch):
    """`run -m` with SPX_DIGEST_PARKED=2: the device's text path (spx_query_text_begin) leaves the digested reads where the
    digestion parked them and the walk takes them by the reads' input offsets -- what a super-batch of hundreds of thousands
    of reads gets by itself.  Every file byte-identical to the oracle harness (compute_ms_pml.cpp:919-938), PML
    with documents and report, and MS pointers (`-M` needs the lengths: concatenated, the same files)."""
    monkeypatch.setenv("SPX_DIGEST_PARKED", "2")
    ref, prefix, seqs, offs, rng = _setup_digested(tmp_path, oracle_mod, 1, 4, 11, seed=66)
    for mode, flags in (("-P", ["-c", "-d", "-w", "50"]), ("-P", []), ("-M", ["-c", "-d", "-w", "50"])):
        _run_both(tmp_path, ref, prefix, "reads.fa", seqs, offs, rng, flags, mode, digest="m")
    monkeypatch.setenv("SPUMONI_SUPER_BATCH", "4000")  # several super-batches, two workers
    monkeypatch.setenv("SPUMONI_GPUS", "0,0")
    _run_both(tmp_path, ref, prefix, "reads.fa", seqs, offs, rng, ["-c", "-d", "-w", "50"], "-P", digest="m")


def test_cli_read_empty_after_digestion_is_fatal_in_order(built, tmp_path, oracle_mod):
    """:926-931 -- a read that digests to nothing (all N / shorter than one window) stops the run
    at that read; what came before is on disk."""
    ref, prefix, seqs, offs, rng = _setup_digested(tmp_path, oracle_mod, 1, 4, 11, seed=71)
    for d in ("gpu", "orc"):
        (tmp_path / d).mkdir(exist_ok=True)
        with open(tmp_path / d / "reads.fa", "w") as f:
            for q in range(25):
                f.write(f">r{q}\n{seqs[offs[q]:offs[q + 1]].tobytes().decode()}\n")
            f.write(">mostly_n\n" + "N" * 90 + "ACGTACGTAC\n>after\n" + "ACGTTGCA" * 10 + "\n")
    r = subprocess.run([HOST_BIN, "run", "-r", ref, "-p", str(tmp_path / "gpu" / "reads.fa"), "-m", "-P", "-c"],
                       capture_output=True)
    o = subprocess.run([ORC_RUN, prefix, str(tmp_path / "orc" / "reads.fa"), "P", "0", "1", "150", "m"], capture_output=True)
    assert r.returncode == 1 and o.returncode == 1
    assert b"mostly_n was empty after digestion" in r.stderr and b"mostly_n was empty after digestion" in o.stderr
    assert r.stdout.endswith(b"\n\n")
    for e in (".pseudo_lengths", ".report"):
        a, b = str(tmp_path / "gpu" / "reads.fa") + e, str(tmp_path / "orc" / "reads.fa") + e
        assert os.path.getsize(b) > 0 and filecmp.cmp(a, b, shallow=False), e


def test_end_to_end_minimizer_index_from_fasta(built, tmp_path):
    """FASTA -> build_index -m (digestion on the device) -> run -m: positives FOUND, nulls not."""
    g1 = synth.random_genome(60_000, seed=31)
    g2 = synth.mutate(g1, seed=32)
    for name, g in (("a.fa", g1), ("b.fa", g2)):
        with open(tmp_path / name, "w") as f:
            f.write(f">{name}\n{g.tobytes().decode()}\n")
    (tmp_path / "list.txt").write_text(f"{tmp_path / 'a.fa'} 1\n{tmp_path / 'b.fa'} 2\n")
    ref = str(tmp_path / "idx" / "pan")
    r = subprocess.run(["python", "-m", "spumoni_amd.build_index", "-l", str(tmp_path / "list.txt"), "-o", ref, "--doc", "-m"],
                       capture_output=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()
    prefix = ref + ".bin"
    dna = np.concatenate([g1, synth.revcomp(g1), g2, synth.revcomp(g2)])
    seqs, offs = synth.sample_reads(dna, 400, 250, seed=5)
    rng = np.random.default_rng(1)
    _run_both(tmp_path, ref, prefix, "reads.fa", seqs, offs, rng, ["-c", "-d", "-w", "50"], "-P", digest="m")
    rep = open(tmp_path / "gpu" / "reads.fa.report").read().splitlines()[1:]
    found = sum("FOUND" in ln and "NOT_PRESENT" not in ln for ln in rep)
    assert 0.3 * len(rep) < found < 0.7 * len(rep)


def test_cli_ms_without_a_text_file(built, tmp_path):
    """`run -M` could not work from the index files alone (the reference reads the text through <ref>.slp).
    Without SPUMONI_TEXT the text is rebuilt from the MS index: .lengths / .pointers / .doc_numbers / .report identical."""
    ref, prefix, seqs, offs, rng = _setup(tmp_path, 47, list(b"ACGT"))
    _run_both(tmp_path, ref, prefix, "reads.fa", seqs, offs, rng, ["-c", "-d"], "-M", give_text=False)


def test_stale_cache_is_not_used(tmp_path):
    """with SPUMONI_CACHE=use a leftover <ref>.pml.spx was loaded whatever index files lay next to it.
    The cache now carries a fingerprint of the files it was written for: rebuild the index under the same
    prefix and `run` flattens the new files (and says why) instead of answering for the old index."""
    ref, prefix, seqs, offs, rng = _setup(tmp_path, 91, list(b"ACGT"))
    a_dir = tmp_path / "gpu"
    a_dir.mkdir()
    _write_fasta(a_dir / "reads.fa", seqs, offs, np.random.default_rng(1))

    def run(policy):
        env = dict(os.environ, SPUMONI_CACHE=policy)
        r = subprocess.run([HOST_BIN, "run", "-r", ref, "-p", str(a_dir / "reads.fa"), "-n", "-P", "-c"],
                           capture_output=True, env=env)
        assert r.returncode == 0, r.stderr.decode()
        return r.stderr.decode(), open(str(a_dir / "reads.fa") + ".pseudo_lengths", "rb").read()

    err, first = run("write")
    assert os.path.exists(prefix + ".pml.spx") and "stale" not in err
    err, again = run("use")
    assert again == first and "stale" not in err
    # another index under the same prefix (other text, other runs): the cache on disk belongs to the old one
    raw2, text2 = cases.real_case(92, 5000, list(b"ACGT"), ndocs=4)
    raw2.write_raw_files(prefix)
    err, third = run("use")
    assert "stale" in err
    os.remove(prefix + ".pml.spx")
    err, fresh = run("off")
    assert third == fresh and third != first


def test_cli_many_small_super_batches_on_the_device(built, tmp_path, monkeypatch):
    """SPUMONI_SUPER_BATCH=3000 characters and SPUMONI_GPUS=0,0: some twenty super-batches through two workers (two index
    replicas on one device), the ordered writer and the report thread, text from the device and formatted on the host:
    the oracle harness's bytes.  And a read that is empty after digestion in the FIRST of many super-batches: nothing of a
    later super-batch may reach the files (the defect the CPU fuzz found in round 3)."""
    monkeypatch.setenv("SPUMONI_SUPER_BATCH", "3000")
    monkeypatch.setenv("SPUMONI_GPUS", "0,0")
    ref, prefix, seqs, offs, rng = _setup(tmp_path, 48, list(b"ACGT"), nreads=400)
    r = _run_both(tmp_path, ref, prefix, "reads.fa", seqs, offs, rng, ["-c", "-d", "-w", "50"], "-P")
    batches = [int(ln.split("(")[1].split()[0]) for ln in r.stderr.decode().splitlines() if "super-batches" in ln]
    assert len(batches) == 2 and sum(batches) >= 10, batches
    _run_both(tmp_path, ref, prefix, "reads.fa", seqs, offs, rng, ["-c", "-d", "-w", "60"], "-M")
    monkeypatch.setenv("SPUMONI_HOST_FORMAT", "1")
    _run_both(tmp_path, ref, prefix, "reads.fa", seqs, offs, rng, ["-c", "-d", "-w", "50"], "-P")
    monkeypatch.delenv("SPUMONI_HOST_FORMAT")
    # fatal in the first super-batch, -a (DNA minimizers): r0 is shorter than one window
    for d in ("gpu", "orc"):
        with open(tmp_path / d / "fatal.fa", "w") as f:
            f.write(">r0\nCTA\n")
            for q in range(1, 200):
                s = seqs[offs[q]: offs[q + 1]].tobytes().decode()
                if len(s) > 40:
                    f.write(f">r{q}\n{s}\n")
    r = subprocess.run([HOST_BIN, "run", "-r", ref, "-p", str(tmp_path / "gpu" / "fatal.fa"), "-a", "-P", "-c", "-K", "3", "-W", "11"], capture_output=True)
    o = subprocess.run([ORC_RUN, prefix, str(tmp_path / "orc" / "fatal.fa"), "P", "0", "1", "150", "a", "--k", "3", "--w", "11"], capture_output=True)
    assert r.returncode == 1 and o.returncode == 1 and b"r0 was empty after digestion" in r.stderr
    for e in (".pseudo_lengths", ".report"):
        assert filecmp.cmp(str(tmp_path / "gpu" / "fatal.fa") + e, str(tmp_path / "orc" / "fatal.fa") + e, shallow=False), e
    assert os.path.getsize(str(tmp_path / "gpu" / "fatal.fa") + ".pseudo_lengths") == 0


@pytest.mark.parametrize("regime", [{"SPUMONI_MAP_MIN": "1"}, {"SPUMONI_MAP_MIN": "1", "SPUMONI_MAP_FACTOR": "0.3"},
                                    {"SPUMONI_MAP_MIN": "1", "SPUMONI_MAP_OUTPUT": "nopin"},
                                    {"SPUMONI_MAP_MIN": "1", "SPUMONI_TRIM_MIN": "0", "SPUMONI_MAP_FACTOR": "4"},
                                    {"SPUMONI_MAP_MIN": "1", "SPUMONI_PIN_SHARE": "1"}])
def test_cli_output_tails_as_memory_on_the_device(built, tmp_path, monkeypatch, regime):
    """Round 5, on the device: the output files' tails prepared as memory and registered with HIP (spx_host_register on a
    MAP_SHARED mapping of the file), the text of every super-batch copied by the device into the file's pages at its place in
    input order -- for these small files too (SPUMONI_MAP_MIN=1); an estimate that is short, so that later super-batches go
    through the slot's buffer and the file's writer thread; mapped without registration (the pool copies the text in).  PML
    and MS with documents and report on two workers and some twenty super-batches: the oracle harness's bytes; the log says
    that the bytes went where the regime says."""
    for k, v in regime.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("SPUMONI_SUPER_BATCH", "3000")
    ref, prefix, seqs, offs, rng = _setup(tmp_path, 49, list(b"ACGT"), nreads=400)
    for mode, extra in (("-P", ["-c", "-d", "-w", "50"]), ("-M", ["-c", "-d", "-w", "60"])):
        r = _run_both(tmp_path, ref, prefix, "reads.fa", seqs, offs, rng, extra, mode)
        err = r.stderr.decode()
        assert "its tail was prepared as memory" in err, err[-1500:]
        line = [ln for ln in err.splitlines() if "output bytes:" in ln][0]
        direct = float(line.split("output bytes:")[1].split("MB")[0])
        staged = float(line.split("pages,")[1].split("MB")[0])
        if "SPUMONI_TRIM_MIN" in regime:
            # (an estimate four times too long: the part of the tail that is not registered with the device is cut off the
            # file beside the run -- EarlyTrim -- while the device writes the registered part)
            cut = [ln for ln in err.splitlines() if "cut off beside the run:" in ln]
            assert cut and float(cut[0].split("cut off beside the run:")[1].split()[0]) > 0, err[-1500:]
            assert direct > 0, line
        elif "SPUMONI_MAP_FACTOR" in regime:
            assert direct > 0 and staged > 0, line
        else:
            assert direct > 0 and staged == 0, line
    assert not [f for f in os.listdir(tmp_path / "gpu") if ".partial." in f]
    # ... and with the reads digested on the device first (-m, -a): the place of a super-batch's text is known only after the
    # digestion; the process must also LEAVE (profiles/r05_cli_e2e_m_hang.txt: a memory pool's destruction once hung here)
    import oracle
    for digest, kw in (("m", ()), ("a", (2, 2))):
        sub = tmp_path / ("dig_" + digest)
        sub.mkdir()
        test_cli_with_minimizer_digestion(None, sub, oracle, digest, kw)


def test_cli_four_workers_on_many_small_super_batches(built, tmp_path, monkeypatch):
    """SPUMONI_GPUS=0,0,0,0: four workers on one device -- one copy of the index, four query contexts --
    pulling some forty super-batches of 1500 characters from one queue, the ordered writer putting them back in
    input order: the oracle harness's bytes in every file, every super-batch accounted for by exactly one worker, and more
    than one worker doing the work (reads are independent: compute_ms_pml.cpp:890-1024)."""
    monkeypatch.setenv("SPUMONI_SUPER_BATCH", "1500")
    monkeypatch.setenv("SPUMONI_GPUS", "0,0,0,0")
    ref, prefix, seqs, offs, rng = _setup(tmp_path, 57, list(b"ACGT"), nreads=500)
    for mode, extra in (("-P", ["-c", "-d", "-w", "50"]), ("-M", ["-c", "-d", "-w", "60"])):
        r = _run_both(tmp_path, ref, prefix, "reads.fa", seqs, offs, rng, extra, mode)
        err = r.stderr.decode()
        batches = [int(ln.split("(")[1].split()[0]) for ln in err.splitlines() if "super-batches" in ln]
        assert len(batches) == 4 and sum(batches) >= 30, batches
        assert sum(1 for b in batches if b > 0) >= 2, batches
        # (round 5: workers that name a device a second time are query contexts over the arrays that are already there --
        # spx_index_clone onto the same device copies nothing; the doubling tree is for OTHER devices)
        assert err.count("a second query context on device 0 (shares the arrays of worker 0)") == 3, err[-2000:]
        assert "index replica on device" not in err


def test_cli_eight_workers_and_eight_feeders_on_a_million_reads(built, tmp_path):
    """What an 8-GPU node's default looks like to the host, on the one device there is (nothing here has run on
    two physical devices): SPUMONI_GPUS=0,0,0,0,0,0,0,0 + SPUMONI_FEEDERS=8 -- eight workers (query contexts over one copy of the
    index) fed by eight parsers from one queue, the ordered completion putting 10^6 reads back in input order (reads are
    independent: compute_ms_pml.cpp:890-1024).  The files must equal the three-worker default's byte for byte, and the run must not
    be slower than it (the window of the reference's timer; a generous bound: boxes jitter)."""
    import re

    raw, text = cases.real_case(61, 300_000, list(b"ACGT"), ndocs=4)
    ref = str(tmp_path / "ref")
    open(ref + ".fa", "w").write(">dummy\n")
    raw.write_raw_files(ref + ".fa")
    from tests.sdsl_files import write_null_db

    write_null_db(ref + ".fa.pmlnulldb", 4.0, [1, 2, 3, 4, 4, 4, 4, 4])
    nreads, m = 1_000_000, 200
    seqs, _ = synth.sample_reads(text, nreads, m, seed=21)
    rows = seqs.reshape(nreads, m)
    reads = tmp_path / "reads.fa"
    with open(reads, "wb") as f:
        for i in range(0, nreads, 100000):
            f.write(b"".join(b">read_%d\n" % j + rows[j].tobytes() + b"\n" for j in range(i, min(nreads, i + 100000))))

    def run(env_extra):
        best, files = None, None
        for _ in range(2):
            r = subprocess.run([HOST_BIN, "run", "-r", ref, "-p", str(reads), "-n", "-P", "-c"], capture_output=True,
                               env=dict(os.environ, SPUMONI_CACHE="off", **env_extra))
            err = r.stderr.decode()
            assert r.returncode == 0 and "finished processing 1000000 reads" in err, err[-2000:]
            secs = [float(x) for x in re.findall(r"done\.\s+\(([0-9.]+) sec\)", err)]
            best = secs[1] if best is None else min(best, secs[1])
            files = {e: open(str(reads) + e, "rb").read() for e in (".pseudo_lengths", ".report")}
        return best, files, err

    t3, f3, _ = run({"SPUMONI_GPUS": "0,0,0"})
    t8, f8, err8 = run({"SPUMONI_GPUS": "0,0,0,0,0,0,0,0", "SPUMONI_FEEDERS": "8"})
    assert f8 == f3 and len(f3[".pseudo_lengths"]) > 400_000_000
    batches = [int(ln.split("(")[1].split()[0]) for ln in err8.splitlines() if "super-batches" in ln]
    assert len(batches) == 8 and sum(1 for b in batches if b > 0) >= 2, batches
    assert "8 feeders" in err8, err8[-1500:]
    assert t8 <= 1.5 * t3 + 0.02, (t8, t3)


def _general_text_case(tmp_path, exe):
    """`run -g -n`: the pattern file is raw bytes, every read ends in \x01 and is named read_<k>; no upper-casing; an empty
    read has an empty values line; what follows the last separator is not a read (compute_ms_pml.cpp:1219-1297).  Both
    modes against the oracle harness; returns the reference prefix."""
    letters = [3, 4, 5, 60, 97, 127, 128, 129, 200, 255]
    ref, prefix, seqs, offs, rng = _setup(tmp_path, 98, letters, n=4000, nreads=60)
    parts = [seqs[offs[q]: offs[q + 1]].tobytes() for q in range(offs.size - 1)]
    data = b"\x01".join(parts[:20]) + b"\x01\x01" + b"\x01".join(parts[20:40]) + b"\x01abc\x00def\x01" + b"trailing text without a separator"
    for d in ("cli", "orc"):
        (tmp_path / d).mkdir()
        (tmp_path / d / "pattern.txt").write_bytes(data)
    for mode, exts in (("P", (".pseudo_lengths",)), ("M", (".lengths", ".pointers"))):
        env = dict(os.environ, SPUMONI_TEXT=prefix + ".rawtext", ASAN_OPTIONS="detect_leaks=0", SPUMONI_SUPER_BATCH="1500")
        r = subprocess.run([exe, "run", "-r", ref, "-p", str(tmp_path / "cli" / "pattern.txt"), "-n", "-g", "-" + mode], capture_output=True, env=env)
        assert r.returncode == 0 and b"Sanitizer" not in r.stderr and b"runtime error" not in r.stderr, r.stderr.decode(errors="replace")[-3000:]
        assert b"finished processing 42 reads" in r.stderr
        o = subprocess.run([ORC_RUN, prefix, str(tmp_path / "orc" / "pattern.txt"), mode, "0", "0", "150", "g", prefix + ".rawtext"], capture_output=True)
        assert o.returncode == 0, o.stderr.decode()
        for e in exts:
            a, b = str(tmp_path / "cli" / "pattern.txt") + e, str(tmp_path / "orc" / "pattern.txt") + e
            assert os.path.getsize(b) > 10000 and filecmp.cmp(a, b, shallow=False), (mode, e)
    return ref


def test_cli_general_text_mode_on_the_device(built, tmp_path):
    _general_text_case(tmp_path, HOST_BIN)
