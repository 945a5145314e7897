"""-m gpu: the HIP path (through the C-ABI) against the CPU oracle, bit for bit."""
import os

import numpy as np
import pytest
import torch

from spumoni_amd import capi, synth
from tests import cases

pytestmark = pytest.mark.gpu

DNA = list(b"ACGT")


@pytest.fixture(scope="module")
def gpu():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    capi.lib()
    return 0


def _compare_all(oracle_mod, raw, text, seqs, offs, ix=None):
    """PML, PML+doc, MS, MS+doc(+lengths) and the classifier, HIP vs oracle."""
    rawc = raw.cpu()
    orc = oracle_mod.OracleIndex.from_raw(rawc)
    ix = ix or capi.Index.from_raw(raw, 0)
    has_docs = raw.doc_start is not None
    # --- PML
    want = orc.pml(seqs, offs)
    got = ix.query_host(capi.SPX_MODE_PML, seqs, offs, classify=(7, 3))
    assert np.array_equal(got["lengths"], want)
    f, a, b, s = oracle_mod.classify(want, offs, 7, 3)
    assert np.array_equal(got["class"]["above"], a)
    assert np.array_equal(got["class"]["below"], b)
    assert np.array_equal(got["class"]["sum_max"], s)
    # classification alone (no per-character values written): the same classes
    only = ix.query_host(capi.SPX_MODE_PML, seqs, offs, want_lengths=False, classify=(7, 3))
    assert "lengths" not in only and np.array_equal(only["class"], got["class"])
    # the 16-bit entry point: same values, half the bytes
    got16 = ix.query_host(capi.SPX_MODE_PML, seqs, offs, classify=(7, 3), bits=16)
    assert got16["lengths"].dtype == np.uint16 and np.array_equal(got16["lengths"], want)
    assert np.array_equal(got16["class"]["above"], a) and np.array_equal(got16["class"]["sum_max"], s)
    if has_docs:
        wl, wd = orc.pml(seqs, offs, want_docs=True)
        got = ix.query_host(capi.SPX_MODE_PML, seqs, offs, want_docs=True)
        assert np.array_equal(got["lengths"], wl)
        assert np.array_equal(got["docs"], wd)
        got16 = ix.query_host(capi.SPX_MODE_PML, seqs, offs, want_docs=True, bits=16)
        assert np.array_equal(got16["lengths"], wl) and np.array_equal(got16["docs"], wd)
    # --- MS
    if raw.ssa is not None:
        w = orc.ms(seqs, offs, want_docs=has_docs, text=text)
        got = ix.query_host(capi.SPX_MODE_MS, seqs, offs, want_lengths=text is not None, want_docs=has_docs,
                            classify=(5, 4) if text is not None else None)
        assert np.array_equal(got["pointers"], w["pointers"])
        if has_docs:
            assert np.array_equal(got["docs"], w["docs"])
        got16 = ix.query_host(capi.SPX_MODE_MS, seqs, offs, want_lengths=text is not None, want_docs=has_docs, bits=16)
        assert np.array_equal(got16["pointers"], w["pointers"])
        if has_docs:
            assert np.array_equal(got16["docs"], w["docs"])
        if text is not None:
            assert np.array_equal(got16["lengths"], w["lengths"])
        if text is not None:
            assert np.array_equal(got["lengths"], w["lengths"])
            f, a, b, s = oracle_mod.classify(w["lengths"], offs, 5, 4)
            assert np.array_equal(got["class"]["above"], a)
            assert np.array_equal(got["class"]["below"], b)
            assert np.array_equal(got["class"]["sum_max"], s)
    st = ix.last_stats()
    return ix, st


@pytest.mark.parametrize(
    "seed,n,letters,extra",
    [
        (1, 300, DNA, [ord("N")]),
        (2, 2000, DNA + [ord("N")], [ord("Z"), 0, 1, 2]),
        (3, 5000, [3, 4, 5, 90, 127, 128, 129, 200, 255], [2, 250]),  # promoted alphabet, bytes >= 128
        (4, 1500, list(range(3, 60)), [1]),
        (5, 64, [ord("A")], [ord("C")]),
    ],
)
@pytest.mark.parametrize("wide_rows", [0, 1])
def test_real_bwt_parity(gpu, oracle_mod, seed, n, letters, extra, wide_rows, monkeypatch):
    if wide_rows:  # the general row encoding (runs of 2^16 and more) on an index that would be compact
        monkeypatch.setenv("SPX_ROWS_WIDE", "1")
    raw, text = cases.real_case(seed, n, letters)
    rng = np.random.default_rng(1000 + seed)
    seqs, offs = cases.reads_mixed(rng, text, letters, 300, 120, extra)
    _compare_all(oracle_mod, raw, text, seqs, offs)


def test_single_read_and_empty_batch(gpu, oracle_mod):
    raw, text = cases.real_case(11, 500, DNA)
    ix = capi.Index.from_raw(raw, 0)
    orc = oracle_mod.OracleIndex.from_raw(raw)
    rd = text[10:90].copy()
    offs = np.array([0, rd.size])
    assert np.array_equal(ix.query_host(capi.SPX_MODE_PML, rd, offs)["lengths"], orc.pml(rd, offs))
    # empty batch / all-empty reads
    out = ix.query_host(capi.SPX_MODE_PML, np.zeros(0, np.uint8), np.array([0]))
    assert out["lengths"].size == 0
    out = ix.query_host(capi.SPX_MODE_PML, np.zeros(0, np.uint8), np.array([0, 0, 0]), classify=(150, 3))
    assert out["lengths"].size == 0 and out["class"].size == 2


@pytest.mark.parametrize("sigma,mean_run,zipf,r", [(4, 6.0, 0.0, 20000), (253, 3.0, 1.0, 50000), (16, 1.0, 0.0, 3000)])
def test_statistical_rlbwt_parity(gpu, oracle_mod, sigma, mean_run, zipf, r):
    letters = DNA if sigma == 4 else None
    raw = synth.statistical_rlbwt(r, sigma, mean_run, seed=sigma, device="cuda", zipf=zipf, letters=letters,
                                  with_samples=True, n_docs=10)
    seqs, offs = synth.simulate_reads(raw, 4000, 60, seed=3, positive_fraction=0.5, f_mis=0.05)
    _, st = _compare_all(oracle_mod, raw, None, seqs.cpu().numpy(), offs.cpu().numpy())
    assert st["steps"] == 4000 * 60


def test_device_resident_query_matches_host_query(gpu, oracle_mod):
    raw = synth.statistical_rlbwt(30000, 253, 4.0, seed=9, device="cuda", zipf=1.0)
    seqs, offs = synth.simulate_reads(raw, 5000, 44, seed=4)
    ix = capi.Index.from_raw(raw, 0)
    d_seqs = capi.pad_seqs(seqs)
    d_len = torch.empty(seqs.numel(), dtype=torch.int32, device="cuda")
    d_cls = torch.empty((offs.numel() - 1, 2), dtype=torch.int64, device="cuda")
    ix.query_device(capi.SPX_MODE_PML, d_seqs, offs, seqs.numel(), d_lengths=d_len, d_class=d_cls, bin_width=150,
                    max_value_thr=5)
    torch.cuda.synchronize()
    orc = oracle_mod.OracleIndex.from_raw(raw.cpu())
    want = orc.pml(seqs.cpu().numpy(), offs.cpu().numpy())
    assert np.array_equal(d_len.cpu().numpy().view(np.uint32), want)
    st = ix.last_stats()
    ost = orc.stats(seqs.cpu().numpy(), offs.cpu().numpy())
    assert (st["steps"], st["jumps"], st["pred_jumps"]) == (ost["steps"], ost["jumps"], ost["pred_jumps"])
    assert st["kernel_ms"] > 0
    # total_chars may be an upper bound (reads digested on the device) ...
    d_len.zero_()
    ix.query_device(capi.SPX_MODE_PML, d_seqs, offs, 3 * seqs.numel() + 1000, d_lengths=d_len, d_class=d_cls,
                    bin_width=150, max_value_thr=5)
    torch.cuda.synchronize()
    assert np.array_equal(d_len.cpu().numpy().view(np.uint32), want)
    ix.last_stats()
    # ... and when it is less than the batch holds: the walk that writes its lengths itself (k_walk_fast) does not
    # need it and is right anyway; the state machine, whose bit-mask scratch it sizes, says so instead of writing
    # past the scratch (tests/test_gpu_old_walk.py runs this test on that kernel)
    d_len.zero_()
    ix.query_device(capi.SPX_MODE_PML, d_seqs, offs, seqs.numel() // 8, d_lengths=d_len, d_class=d_cls, bin_width=150,
                    max_value_thr=5)
    torch.cuda.synchronize()
    if os.environ.get("SPX_OLD_WALK"):
        with pytest.raises(capi.SpxError, match="total_chars"):
            ix.last_stats()
    else:
        assert np.array_equal(d_len.cpu().numpy().view(np.uint32), want)
        ix.last_stats()
    ix.query_device(capi.SPX_MODE_PML, d_seqs, offs, seqs.numel(), d_lengths=d_len, d_class=d_cls, bin_width=150,
                    max_value_thr=5)
    torch.cuda.synchronize()
    assert np.array_equal(d_len.cpu().numpy().view(np.uint32), want) and ix.last_stats()["steps"] == ost["steps"]


def test_raw_file_loader(gpu, oracle_mod, tmp_path):
    raw, text = cases.real_case(21, 3000, DNA)
    prefix = str(tmp_path / "idx")
    raw.write_raw_files(prefix)
    rng = np.random.default_rng(5)
    seqs, offs = cases.reads_mixed(rng, text, DNA, 100, 100)
    orc = oracle_mod.OracleIndex.from_raw(raw)
    for mode in (capi.SPX_MODE_PML, capi.SPX_MODE_MS):
        ix = capi.Index.load_raw(prefix, mode, 0)
        assert (ix.n, ix.r) == (raw.n, raw.r)
        got = ix.query_host(mode, seqs, offs, want_lengths=(mode == capi.SPX_MODE_PML))
        if mode == capi.SPX_MODE_PML:
            assert np.array_equal(got["lengths"], orc.pml(seqs, offs))
        else:
            assert np.array_equal(got["pointers"], orc.ms(seqs, offs)["pointers"])


def test_invalid_index_is_rejected(gpu):
    raw, _ = cases.real_case(31, 400, DNA)
    bad = raw.cpu()
    bad.lens = bad.lens.clone()
    bad.lens[3] = 0
    with pytest.raises(capi.SpxError):
        capi.Index.from_raw(bad, 0)


def test_config1_ecoli_scale(gpu, oracle_mod):
    """BASELINE config[0] shape: single 4.64 Mbp genome (+revcomp), 150 bp reads, PML -c."""
    g = synth.random_genome(4_641_652, seed=1)
    text, doc_lengths = synth.pangenome_text([g])
    raw = synth.index_from_text(torch.from_numpy(text).cuda(), doc_lengths=doc_lengths, with_samples=False)
    seqs, offs = synth.sample_reads(text, 100_000, 150, seed=11)  # the declared 100 000 x 150 bp
    ix = capi.Index.from_raw(raw, 0)
    got = ix.query_host(capi.SPX_MODE_PML, seqs, offs, classify=(150, 3 + 4))
    orc = oracle_mod.OracleIndex.from_raw(raw.cpu())
    want = orc.pml(seqs, offs)
    assert np.array_equal(got["lengths"], want)
    # the .report's columns (compute_ms_pml.cpp:969-995): bins above / below the threshold, the sum their mean is printed from
    f, a, b, s = oracle_mod.classify(want, offs, 150, 7)
    assert np.array_equal(got["class"]["above"], a) and np.array_equal(got["class"]["below"], b)
    assert np.array_equal(got["class"]["sum_max"], s)
    assert np.array_equal(2 * got["class"]["above"].astype(np.int64) > got["class"]["above"].astype(np.int64) + got["class"]["below"], f.astype(bool))
    # sampled reads are FOUND, reversed (null) reads are not: the classifier separates them
    assert 0.3 < f.mean() < 0.7


def _adaptive_reads(orc, letters, nreads, length, rng, p_head=0.7):
    """Reads built by following the oracle's own walk: with probability p_head the next character
    is the head of the run the walk sits on (for bytes >= 128 that is the Appendix-C1 situation)."""
    n = orc.n
    out = np.zeros((nreads, length), dtype=np.uint8)
    for q in range(nreads):
        pos = n - 1
        for i in range(length):
            if pos < n and rng.random() < p_head and orc.at(pos) > 1:
                c = orc.at(pos)
            else:
                c = int(letters[rng.integers(0, len(letters))])
            out[q, length - 1 - i] = c
            nc = orc.rank(n, c)
            if nc == 0:
                pass
            elif pos < n and orc.at(pos) == c and c < 128:
                pass
            else:
                rnk = orc.rank(pos, c)
                thr = n + 1
                nxt = pos
                if rnk < nc:
                    j = orc.select(rnk, c)
                    thr = orc.threshold(orc.run_of_position(j))
                    nxt = j
                if pos < thr:
                    nxt = orc.select(rnk - 1, c)
                pos = nxt
            pos = orc.LF(pos, c)
    offs = np.arange(nreads + 1, dtype=np.int64) * length
    return out.reshape(-1), offs


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_inconsistent_thresholds_general_path(gpu, oracle_mod, seed):
    """Thresholds anywhere in [1, n] (not between the neighbouring same-letter runs): every step is
    still defined upstream, and bytes >= 128 that sit on their own run now take the threshold
    branch for real (Appendix C1 with pos < thr: select(rnk-1) inside the run / previous run)."""
    rng = np.random.default_rng(seed)
    letters = list(range(100, 140))
    raw = synth.statistical_rlbwt(3000, 40, 3.0, seed=seed, letters=letters, with_samples=True, n_docs=6)
    thr = raw.thr.clone()
    nz = thr > 0  # first run of a letter keeps threshold 0 (thr_bv semantics)
    rnd = torch.from_numpy(rng.integers(1, raw.n + 1, size=raw.r))
    raw.thr = torch.where(nz, rnd, thr)
    orc = oracle_mod.OracleIndex.from_raw(raw)
    seqs, offs = _adaptive_reads(orc, letters, 300, 40, rng)
    hi_on_head = int((seqs >= 128).sum())
    assert hi_on_head > 1000
    _, st = _compare_all(oracle_mod, raw, None, seqs, offs)
    assert st["pred_jumps"] > 0


def test_degenerate_indexes(gpu, oracle_mod):
    """One-run and two-run indexes, reads made of the terminator byte / absent letters."""
    for heads, lens in (([0], [1]), ([65, 0], [3, 1]), ([0, 65], [1, 5])):
        raw = synth.RawIndex(heads=torch.tensor(heads, dtype=torch.uint8), lens=torch.tensor(lens, dtype=torch.int64),
                             thr=torch.zeros(len(heads), dtype=torch.int64), n=sum(lens),
                             ssa=torch.zeros(len(heads), dtype=torch.int64), esa=torch.zeros(len(heads), dtype=torch.int64),
                             doc_start=torch.zeros(len(heads), dtype=torch.int64),
                             doc_end=torch.zeros(len(heads), dtype=torch.int64))
        seqs = np.array([65, 65, 1, 66, 65, 0, 1, 1, 65], dtype=np.uint8)
        offs = np.array([0, 4, 4, 9])
        _compare_all(oracle_mod, raw, None, seqs, offs)


def test_limits_are_enforced(gpu):
    base = dict(heads=torch.tensor([0, 65, 67], dtype=torch.uint8), thr=torch.zeros(3, dtype=torch.int64))
    # BWT longer than the 40-bit position field
    with pytest.raises(capi.SpxError):
        capi.Index.from_raw(synth.RawIndex(lens=torch.tensor([1, 1 << 39, 1 << 39], dtype=torch.int64), n=(1 << 40) + 1, **base), 0)
    # document id that does not fit 16 bits
    with pytest.raises(capi.SpxError):
        capi.Index.from_raw(synth.RawIndex(lens=torch.tensor([1, 2, 3], dtype=torch.int64), n=6,
                                           doc_start=torch.tensor([0, 70000, 1]), doc_end=torch.tensor([0, 1, 1]), **base), 0)
    # MS query on a PML-only index, docs on an index without a document array
    ix = capi.Index.from_raw(synth.RawIndex(lens=torch.tensor([1, 2, 3], dtype=torch.int64), n=6, **base), 0)
    rd, offs = np.array([65, 67], dtype=np.uint8), np.array([0, 2])
    with pytest.raises(capi.SpxError):
        ix.query_host(capi.SPX_MODE_MS, rd, offs)
    with pytest.raises(capi.SpxError):
        ix.query_host(capi.SPX_MODE_PML, rd, offs, want_docs=True)


@pytest.mark.parametrize("bshift,wide_rows,all_esc", [(0, 0, 0), (2, 1, 0), (5, 0, 0), (9, 1, 0), (0, 0, 1), (3, 1, 1),
                                                      ("0.05:1.0", 0, 0), ("0.8:0.7", 1, 0), ("3:0.5", 0, 1),
                                                      ("12:0.0", 0, 0)])
def test_every_directory_block_size(gpu, oracle_mod, bshift, wide_rows, all_esc, monkeypatch):
    """The fat table's geometry is chosen from the free memory; force it -- one block size for all
    letters from 1 to 512 runs, or per-letter block sizes at a given table density ("slots per
    run:alpha") -- so that the direct answer, the next-slot answer, the one-window and the
    multi-window directory scans all run (with both row encodings); all_esc marks every 16-byte fat
    digest as not holding its row, so that every jump goes through fat_j and the full JumpRow."""
    if isinstance(bshift, str):
        spr, alpha = bshift.split(":")
        monkeypatch.setenv("SPX_FAT_SLOTS_PER_RUN", spr)
        monkeypatch.setenv("SPX_FAT_ALPHA", alpha)
    else:
        monkeypatch.setenv("SPX_FAT_BSHIFT", str(bshift))
    if all_esc:
        monkeypatch.setenv("SPX_FAT_ALL_ESC", "1")
    if wide_rows:
        monkeypatch.setenv("SPX_ROWS_WIDE", "1")
    for seed, letters in ((61, DNA), (62, [3, 4, 5, 90, 127, 128, 129, 200, 255])):
        raw, text = cases.real_case(seed, 6000, letters)
        rng = np.random.default_rng(seed)
        seqs, offs = cases.reads_mixed(rng, text, letters, 250, 100, [ord("N")])
        _, st = _compare_all(oracle_mod, raw, text, seqs, offs)
    raw = synth.statistical_rlbwt(30000, 16, 2.0, seed=7, device="cuda", with_samples=True, n_docs=5)
    seqs, offs = synth.simulate_reads(raw, 3000, 50, seed=8, positive_fraction=0.3)
    _compare_all(oracle_mod, raw, None, seqs.cpu().numpy(), offs.cpu().numpy())


def test_long_runs_and_far_thresholds(gpu, oracle_mod):
    """Runs of 2^16 and more (the index then keeps the general row encoding instead of the compact
    one, and offsets of 2^16 and more do not fit the 16-byte fat digest) and a letter whose two
    runs -- and so a threshold and its run -- lie 1.5 * 2^20 runs apart (distance does not fit):
    those slots take the escape path, the rest of the same index the 16-byte path."""
    rng = np.random.default_rng(5)
    acg = np.frombuffer(b"ACG", dtype=np.uint8)
    # (1) long runs among short ones
    r = 4000
    idx = rng.integers(0, 3, size=r)
    for i in range(1, r):
        if idx[i] == idx[i - 1]:
            idx[i] = (idx[i] + 1) % 3
    heads = acg[idx].copy()
    lens = rng.integers(1, 6, size=r).astype(np.int64)
    big = rng.random(r) < 0.05
    lens[big] = rng.integers(1 << 16, 3 << 16, size=int(big.sum()))
    heads[r // 2], lens[r // 2] = 0, 1
    raw = synth.raw_from_runs(torch.from_numpy(heads), torch.from_numpy(lens), 3, with_samples=True, n_docs=4)
    seqs, offs = synth.simulate_reads(raw, 2000, 60, seed=9, positive_fraction=0.5)
    _compare_all(oracle_mod, raw, None, seqs.cpu().numpy(), offs.cpu().numpy())
    # (2) a letter with two runs 1.5 * 2^20 runs apart
    r = (1 << 21) + 1000
    # neighbours differ by construction: each head is its predecessor + 1 or + 2 (mod 3).  (Repairing equal
    # neighbours of an i.i.d. draw in place pushes the conflicts along for tens of thousands of passes over the
    # 2 * 10^6 runs: 220 s of this test's 231 s on the GPU box, tools/slow_test_probe.py)
    idx = np.cumsum(rng.integers(1, 3, size=r)) % 3
    assert not (idx[1:] == idx[:-1]).any()
    heads = acg[idx].copy()
    lens = rng.integers(1, 4, size=r).astype(np.int64)
    heads[0], lens[0] = 0, 1
    heads[500] = heads[500 + (3 << 19)] = ord("T")
    for seed in (1, 2, 3):  # the threshold is drawn uniformly between the two T runs
        raw = synth.raw_from_runs(torch.from_numpy(heads), torch.from_numpy(lens), seed, with_samples=True, n_docs=3)
        seqs, offs = synth.simulate_reads(raw, 20000, 40, seed=seed, positive_fraction=0.5)
        seqs = seqs.cpu().numpy().copy()
        seqs[rng.random(seqs.size) < 0.05] = ord("T")  # jumps to the rare letter from everywhere
        _compare_all(oracle_mod, raw, None, seqs, offs.cpu().numpy())


def test_balanced_pieces_bound_the_walk_past_the_landing_row(gpu, oracle_mod, monkeypatch):
    """Heavy-tailed run lengths (Pareto, longest run 7871 among runs of 1-2): the LF image of a long run covers up to
    1393 runs, and a step out of it walks on row by row from the fifth (42 gathers per step for positions drawn
    uniformly, tools/ff_model.py).  The flatten step cuts such runs into pieces whose images cover at most 8 runs
    (spx_layout.h: for_each_piece; the pass is repeated on its own output): some 6 % more rows, the same answers in
    every mode, and a fraction of the row gathers; SPX_BALANCE_SPAN=0 is the layout without it."""
    rng = np.random.default_rng(8)
    r = 1 << 16
    idx = np.cumsum(rng.integers(1, 4, size=r)) % 4
    lens = np.minimum((rng.pareto(1.2, size=r) + 1).astype(np.int64), 1 << 18)
    heads = np.frombuffer(b"ACGT", dtype=np.uint8)[idx].copy()
    heads[r // 2], lens[r // 2] = 0, 1
    assert 2048 <= lens.max() < 65536  # (no piece is cut for its length here)
    raw = synth.raw_from_runs(torch.from_numpy(heads), torch.from_numpy(lens), 4, with_samples=True, n_docs=3)
    seqs, offs = synth.simulate_reads(raw, 3000, 80, seed=2, positive_fraction=0.7)
    s, o = seqs.cpu().numpy(), offs.cpu().numpy()
    ix, st = _compare_all(oracle_mod, raw, None, s, o)
    d = ix.describe()
    monkeypatch.setenv("SPX_BALANCE_SPAN", "0")
    ix0 = capi.Index.from_raw(raw, 0)
    _, st0 = _compare_all(oracle_mod, raw, None, s, o, ix=ix0)
    d0 = ix0.describe()
    assert d["r"] == d0["r"] == raw.r == d0["flat_runs"] and d["compact_rows"] == d0["compact_rows"] == 1
    assert d0["flat_runs"] < d["flat_runs"] <= d0["flat_runs"] + raw.r // 8
    assert (st["steps"], st["jumps"]) == (st0["steps"], st0["jumps"])
    assert 3 * st["row_loads"] < st0["row_loads"], (st, st0)


def test_16_bit_outputs_refuse_long_reads(gpu):
    raw, text = cases.real_case(3, 400, DNA)
    ix = capi.Index.from_raw(raw, 0)
    seqs = np.full(70_000, ord("A"), dtype=np.uint8)
    offs = np.array([0, 70_000])
    with pytest.raises(capi.SpxError):
        ix.query_host(capi.SPX_MODE_PML, seqs, offs, bits=16)
    assert ix.query_host(capi.SPX_MODE_PML, seqs, offs)["lengths"].size == 70_000


@pytest.mark.parametrize("bits", [16, 32])
def test_lengths_from_reset_bits_edge_shapes(gpu, oracle_mod, bits):
    """The plain PML walk hands its lengths over as one bit per character ("the length was reset here") and
    k_expand_lengths writes them out (compute_ms_pml.cpp:249-250, 266-276 read from the other side).  Read lengths
    around every boundary of that encoding (64-bit words, 128-character pairs, groups of 8 outputs, reads that start
    at any alignment), exact substrings (no reset for hundreds of characters: the scan for the next set bit crosses
    words), substrings with one error, random reads, absent letters, empty reads."""
    raw, text = cases.real_case(77, 30000, DNA, ndocs=2)
    rng = np.random.default_rng(77)
    lens = [0, 1, 2, 7, 8, 9, 15, 16, 17, 63, 64, 65, 120, 127, 128, 129, 130, 191, 192, 193, 255, 256, 257, 383, 384,
            385, 1000, 1023, 1024, 1025, 5000]
    reads = []
    for rep in range(3):
        for m in lens:
            s = int(rng.integers(0, text.size - m)) if m else 0
            exact = text[s:s + m].copy()
            reads.append(exact)
            one = exact.copy()
            if m:
                one[rng.integers(0, m)] = ord("N")  # absent letter: the length restarts at 0 there
            reads.append(one)
            reads.append(np.asarray(DNA, dtype=np.uint8)[rng.integers(0, 4, size=m)])
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    offs = np.concatenate([[0], np.cumsum([r.size for r in reads])]).astype(np.int64)
    seqs = np.concatenate(reads)
    orc = oracle_mod.OracleIndex.from_raw(raw.cpu())
    want = orc.pml(seqs, offs)
    ix = capi.Index.from_raw(raw, 0)
    ix.set_option("chunk_mode", 1)  # never: this is about the plain walk
    got = ix.query_host(capi.SPX_MODE_PML, seqs, offs, classify=(50, 10), bits=bits)
    assert np.array_equal(got["lengths"], want)
    # long reset-free stretches are really there (otherwise the word-crossing scan is not exercised)
    assert want.max() > 1000
    _, a, b, s = oracle_mod.classify(want, offs, 50, 10)
    assert np.array_equal(got["class"]["above"], a) and np.array_equal(got["class"]["sum_max"], s)
    wl, wd = orc.pml(seqs, offs, want_docs=True)
    gd = ix.query_host(capi.SPX_MODE_PML, seqs, offs, want_docs=True, bits=bits)
    assert np.array_equal(gd["lengths"], wl) and np.array_equal(gd["docs"], wd)


@pytest.mark.parametrize("lpw", [1, 5, 33, 64])
@pytest.mark.parametrize("waves", [4, 28])
def test_lane_and_occupancy_knobs_do_not_change_results(gpu, oracle_mod, lpw, waves):
    """`lanes_per_wave` (1 = the "one wavefront owns one read" mapping of SURVEY 7.1) and `waves_per_cu`: launch
    geometry only.  With fewer active lanes per wavefront the idle lanes still take part in the wavefront-wide
    stores of the lengths and MS pointers (k_walk_fast): every mode against the oracle."""
    raw, text = cases.real_case(81, 7000, [3, 4, 5, 90, 127, 128, 129, 200, 255], ndocs=4)
    rng = np.random.default_rng(8)
    seqs, offs = cases.reads_mixed(rng, text, [3, 4, 5, 90, 127, 128, 129, 200, 255], 150, 300, [2])
    ix = capi.Index.from_raw(raw, 0)
    ix.set_option("lanes_per_wave", lpw)
    ix.set_option("waves_per_cu", waves)
    _compare_all(oracle_mod, raw, text, seqs, offs, ix=ix)


@pytest.mark.parametrize("slots", ["0.3", "2.0", "12"])
def test_fat_slots_with_their_landing_row(gpu, oracle_mod, monkeypatch, slots):
    """SPX_FAT_LROW=1 (experiment knob): a PML-only index with compact rows keeps, in the second half of a
    32-byte fat slot, the row of the run the slot's jump lands in, and the step after such a jump goes on from it without
    a landing gather.  The same values as the oracle's (compute_ms_pml.cpp:238-286) on a statistical index, on a real BWT
    with bytes >= 128 and absent letters, through the chunked walk, and from the flat-layout cache."""
    monkeypatch.setenv("SPX_FAT_LROW", "1")
    monkeypatch.setenv("SPX_FAT_SLOTS_PER_RUN", slots)
    raw = synth.statistical_rlbwt(40000, 60, 3.0, seed=11, zipf=1.0)
    assert raw.ssa is None and raw.doc_start is None
    seqs, offs = synth.simulate_reads(raw, 3000, 70, seed=5, positive_fraction=0.6, warmup=2)
    seqs = seqs.cpu().numpy().copy()
    seqs[::97] = 2  # a letter the index does not have
    ix, st = _compare_all(oracle_mod, raw, None, seqs, offs.cpu().numpy())
    d = ix.describe()
    assert d["fat_stride"] == 32 and d["has_samples"] == 0 and d["compact_rows"] == 1
    # long reads through the chunked walk (pass 1 is k_walk_fast too)
    ls, lo = synth.simulate_reads(raw, 30, 1200, seed=6, positive_fraction=0.7, warmup=2)
    ix.set_option("chunk_mode", 2)
    got = ix.query_host(capi.SPX_MODE_PML, ls.cpu().numpy(), lo.cpu().numpy(), classify=(150, 5))
    assert ix.last_chunk_stats()["chunk_len"] > 0
    assert np.array_equal(got["lengths"], oracle_mod.OracleIndex.from_raw(raw.cpu()).pml(ls.cpu().numpy(), lo.cpu().numpy()))
    ix.close()
    # a real BWT with bytes >= 128 (the stay-put jump) and reads with absent letters; samples / documents dropped
    from tests import cases

    raw2, _ = cases.real_case(3, 5000, [3, 4, 5, 90, 127, 128, 129, 200, 255])
    raw2.ssa = raw2.esa = raw2.doc_start = raw2.doc_end = None
    rng = np.random.default_rng(8)
    s2, o2 = cases.reads_mixed(rng, None, [3, 4, 5, 90, 127, 128, 129, 200, 255], 400, 120, [2, 250])
    ix2, _ = _compare_all(oracle_mod, raw2, None, s2, o2)
    assert ix2.describe()["fat_stride"] == 32
