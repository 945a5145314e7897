"""-m gpu: the chunked walk of long-read batches (BASELINE config 5) against the oracle.

A batch with fewer reads than the GPU has lanes is cut into chunks that are walked speculatively
from the default state, joined by carrying the true walk over every seam until it meets the
speculative one, and patched.  The result must be the plain walk's, bit for bit:
PML, PML + doc, MS pointers (+ doc, + lengths), 16- and 32-bit outputs, the classifier -- for every
chunk size, for reads that end inside / on / next to chunk boundaries, for reads that never jump
(one long match: the seams only close at the read's end or not at all -> the fallback)."""
import numpy as np
import pytest
import torch

from spumoni_amd import capi, synth
from tests import cases
from tests.test_gpu_parity import _compare_all

pytestmark = pytest.mark.gpu

DNA = list(b"ACGT")


def _chunked(ix, shift):
    """shift < 20: chunks of 2^shift characters; otherwise the chunk size itself (any multiple of 16)."""
    ix.set_option("chunk_mode", 2)  # always
    if shift < 20:
        ix.set_option("chunk_shift", shift)
    else:
        ix.set_option("chunk_len", shift)
    return ix


@pytest.mark.parametrize("shift", [5, 6, 8, 48, 112])
@pytest.mark.parametrize("wide_rows", [0, 1])
def test_real_index_ragged_reads(oracle_mod, shift, wide_rows, monkeypatch):
    """Ragged reads (0 .. 700 characters, some empty) on a real BWT with documents: every read end falls
    somewhere else relative to the chunk grid.  wide_rows: the general row encoding is not chunked --
    the request must fall back to the plain walk and still be right."""
    if wide_rows:
        monkeypatch.setenv("SPX_ROWS_WIDE", "1")
    for seed, letters, extra in ((71, DNA, [ord("N")]), (72, [3, 4, 5, 90, 127, 128, 129, 200, 255], [2, 250])):
        raw, text = cases.real_case(seed, 9000, letters, ndocs=4)
        rng = np.random.default_rng(seed)
        seqs, offs = cases.reads_mixed(rng, text, letters, 160, 700, extra)
        ix = _chunked(capi.Index.from_raw(raw, 0), shift)
        _compare_all(oracle_mod, raw, text, seqs, offs, ix=ix)


@pytest.mark.parametrize("shift", [5, 7, 176])
def test_statistical_index_long_reads(oracle_mod, shift):
    """config 5 shape, small: promoted alphabet (bytes >= 128 take the quirk paths), 300 x 2200."""
    raw = synth.statistical_rlbwt(1 << 16, 253, 8.0, seed=6, device="cuda", zipf=1.0, with_samples=True, n_docs=10)
    seqs, offs = synth.simulate_reads(raw, 300, 2200, seed=16)
    ix = _chunked(capi.Index.from_raw(raw, 0), shift)
    _compare_all(oracle_mod, raw, None, seqs.cpu().numpy(), offs.cpu().numpy(), ix=ix)
    st, cs = ix.last_stats(), ix.last_chunk_stats()
    assert cs["chunk_len"] == (1 << shift if shift < 20 else shift) and cs["rewalked_chars"] > 0
    # characters walked a second time to join chunks are not steps; reads that fell back are walked again in full
    assert 300 * 2200 <= st["steps"] <= 300 * 2200 + cs["fallback_reads"] * 2200
    if shift >= 7:
        assert cs["fallback_reads"] == 0  # chunks of 128 and more: a seam left open closes in the next round


def test_reads_that_never_jump_fall_back(oracle_mod):
    """A read that is one long exact match never resets its counters, and a speculative walk that sits
    in another copy of the repeat never meets the true one: seams stay open, the read is walked again
    plainly.  Also: absent letters everywhere (every step resets), and reads of one repeated letter."""
    rng = np.random.default_rng(9)
    unit = np.frombuffer(b"ACGTTGCAAGGCTTAACCGT", dtype=np.uint8)
    text = np.tile(unit, 400)  # 8000 characters, period 20: every long substring occurs ~400 times
    raw = synth.index_from_text(torch.from_numpy(text.copy()), doc_lengths=[3000, 5000])
    reads = [text[7:3007], text[100:1500], np.full(900, ord("N"), dtype=np.uint8), np.full(1000, ord("A"), dtype=np.uint8),
             text[13:2013].copy()]
    reads[4][::97] = ord("T")  # the same with sparse mismatches
    offs = np.concatenate([[0], np.cumsum([r.size for r in reads])]).astype(np.int64)
    seqs = np.concatenate(reads)
    for shift in (5, 6, 9, 96):
        ix = _chunked(capi.Index.from_raw(raw, 0), shift)
        _compare_all(oracle_mod, raw, text, seqs, offs, ix=ix)


# (a longer sweep: SPX_FUZZ_FIRST / SPX_FUZZ_SEEDS, as in test_gpu_fuzz.py)
_FIRST = int(__import__("os").environ.get("SPX_FUZZ_FIRST", "0"))
_COUNT = int(__import__("os").environ.get("SPX_FUZZ_SEEDS", "12")) if "SPX_FUZZ_FIRST" in __import__("os").environ else 12


@pytest.mark.parametrize("seed", range(_FIRST, _FIRST + _COUNT))
def test_chunk_boundary_fuzz(oracle_mod, seed):
    """Random index shapes x chunk sizes x read lengths around multiples of the chunk size."""
    rng = np.random.default_rng(4000 + seed)
    sigma = int(rng.choice([4, 5, 17, 200]))
    lo = int(rng.choice([3, 60, 120]))
    letters = sorted(rng.choice(np.arange(lo, 256), size=min(sigma, 256 - lo), replace=False).tolist())
    r = int(rng.choice([400, 3000, 20000]))
    raw = synth.statistical_rlbwt(r, len(letters), float(rng.choice([1.0, 2.0, 8.0, 40.0])), seed=seed, letters=letters,
                                  zipf=float(rng.choice([0.0, 1.0])), with_samples=True, n_docs=int(rng.choice([1, 3, 200])))
    if seed % 3 == 2:  # thresholds anywhere (Appendix C1 general path)
        nz = raw.thr > 0
        raw.thr = torch.where(nz, torch.from_numpy(rng.integers(1, raw.n + 1, size=raw.r)), raw.thr)
    shift = int(rng.choice([5, 6, 7, 48, 80]))
    L = 1 << shift if shift < 20 else shift
    lens = []
    for _ in range(int(rng.choice([3, 40]))):
        lens.append(int(rng.choice([1, L - 1, L, L + 1, 2 * L, 3 * L - 1, 5 * L + 7, 16, 17, 0])) + int(rng.integers(0, 3)) * L)
    nreads = len(lens)
    m = max(lens) if max(lens) > 0 else 1
    pool, _ = synth.simulate_reads(raw, nreads, m, seed=seed, positive_fraction=float(rng.choice([0.3, 1.0])),
                                   f_mis=float(rng.choice([0.0, 0.02, 0.2])))
    pool = pool.cpu().numpy().reshape(nreads, m)
    absent = [c for c in range(2, 256) if c not in letters][:1]
    reads = []
    for q, ln in enumerate(lens):
        rd = pool[q, m - ln:].copy() if ln else np.zeros(0, dtype=np.uint8)
        if absent and seed % 2 == 0 and ln:
            rd[rng.random(ln) < 0.02] = absent[0]
        reads.append(rd)
    offs = np.concatenate([[0], np.cumsum([x.size for x in reads])]).astype(np.int64)
    seqs = np.concatenate(reads) if offs[-1] else np.zeros(0, dtype=np.uint8)
    ix = _chunked(capi.Index.from_raw(raw, 0), shift)
    _compare_all(oracle_mod, raw, None, seqs, offs, ix=ix)


def test_automatic_choice_and_device_entry_point(oracle_mod):
    """Nothing forced: a batch of few long reads is chunked, a batch of many short reads is not, and both
    equal the oracle; the device entry point with the classifier."""
    raw = synth.statistical_rlbwt(1 << 18, 253, 8.0, seed=6, device="cuda", zipf=1.0)
    ix = capi.Index.from_raw(raw, 0)
    orc = oracle_mod.OracleIndex.from_raw(raw.cpu())
    for nreads, m in ((500, 3000), (20000, 44)):
        seqs, offs = synth.simulate_reads(raw, nreads, m, seed=nreads)
        d_seqs = capi.pad_seqs(seqs)
        d_len = torch.empty(seqs.numel() + 8, dtype=torch.int32, device="cuda")
        d_cls = torch.empty((nreads, 2), dtype=torch.int64, device="cuda")
        ix.query_device(capi.SPX_MODE_PML, d_seqs, offs, seqs.numel(), d_lengths=d_len, d_class=d_cls, bin_width=150,
                        max_value_thr=5)
        torch.cuda.synchronize()
        ix.last_stats()
        want = orc.pml(seqs.cpu().numpy(), offs.cpu().numpy())
        assert np.array_equal(d_len[: seqs.numel()].cpu().numpy().view(np.uint32), want)
        f, a, b, s = oracle_mod.classify(want, offs.cpu().numpy(), 150, 5)
        c32 = d_cls.view(torch.int32).view(nreads, 4).cpu().numpy()
        assert np.array_equal(c32[:, 2].view(np.uint32), a) and np.array_equal(c32[:, 3].view(np.uint32), b)
        assert np.array_equal(d_cls[:, 0].cpu().numpy().view(np.uint64), s)


def test_pipelined_host_batch_pieces_take_the_chunked_walk(oracle_mod):
    """A host batch of >= 2^18 reads and >= 64 MB runs as a pipeline of (growing) pieces whose offsets stay absolute: every
    piece takes the chunked walk with its scratch (flags, checkpoints) indexed from the piece's first character.  Same
    values as the plain walk of the whole batch, and as the oracle on a sample."""
    raw = synth.statistical_rlbwt(1 << 18, 60, 6.0, seed=8, device="cuda", zipf=1.0, with_samples=True, n_docs=7)
    nreads = 8 * 32768 + 5
    rng = np.random.default_rng(2)
    lens = rng.integers(400, 700, size=nreads)
    lens[::1000] = 0  # some empty reads
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    pool, _ = synth.simulate_reads(raw, 4096, 700, seed=9, f_mis=0.05, warmup=3)
    pool = pool.cpu().numpy().reshape(4096, 700)
    seqs = np.concatenate([pool[q % 4096, 700 - l:] for q, l in enumerate(lens)])
    assert seqs.size >= (64 << 20)
    ix = capi.Index.from_raw(raw, 0)
    ix.set_option("chunk_mode", 1)
    want = ix.query_host(capi.SPX_MODE_PML, seqs, offs, want_docs=True, classify=(150, 5))
    want_ms = ix.query_host(capi.SPX_MODE_MS, seqs, offs, want_lengths=False, want_docs=True)
    ix.set_option("chunk_mode", 2)
    ix.set_option("chunk_len", 96)
    for bits in (32, 16):
        got = ix.query_host(capi.SPX_MODE_PML, seqs, offs, want_docs=True, classify=(150, 5), bits=bits)
        assert ix.last_chunk_stats()["chunk_len"] == 96
        assert np.array_equal(got["lengths"], want["lengths"]) and np.array_equal(got["docs"], want["docs"])
        assert np.array_equal(got["class"], want["class"])
    got_ms = ix.query_host(capi.SPX_MODE_MS, seqs, offs, want_lengths=False, want_docs=True)
    assert np.array_equal(got_ms["pointers"], want_ms["pointers"]) and np.array_equal(got_ms["docs"], want_ms["docs"])
    ns = 300
    orc = oracle_mod.OracleIndex.from_raw(raw.cpu())
    wl, wd = orc.pml(seqs[: offs[ns]], offs[: ns + 1], want_docs=True)
    assert np.array_equal(want["lengths"][: offs[ns]], wl) and np.array_equal(want["docs"][: offs[ns]], wd)


@pytest.mark.parametrize("shift_by", [1, 16, 37, 4099])
@pytest.mark.parametrize("mode_doc", [(0, False), (0, True), (1, True)])
def test_device_path_with_nonzero_first_offset(oracle_mod, shift_by, mode_doc):
    """spx_query_batch_device with d_offsets[0] != 0 (the header allows it: total_chars =
    d_offsets[nreads] - d_offsets[0]) on the chunked walk: the per-character scratch (flag bytes,
    checkpoints) is sized by total_chars, so the kernels index it relative to offs[0]."""
    mode, doc = mode_doc
    raw = synth.statistical_rlbwt(1 << 15, 253, 6.0, seed=8, device="cuda", zipf=1.0, with_samples=True, n_docs=6)
    seqs, offs = synth.simulate_reads(raw, 60, 1700, seed=18, warmup=2)
    total = int(seqs.numel())
    orc = oracle_mod.OracleIndex.from_raw(raw.cpu())
    hs, ho = seqs.cpu().numpy(), offs.cpu().numpy()
    ix = _chunked(capi.Index.from_raw(raw, 0), 7)
    # the same reads, preceded by `shift_by` bytes that belong to no read
    d_seqs = torch.zeros(shift_by + total + 64, dtype=torch.uint8, device="cuda")
    d_seqs[shift_by: shift_by + total] = seqs
    d_offs = (offs + shift_by).contiguous()
    d_len = torch.full((shift_by + total + 8,), -1, dtype=torch.int32, device="cuda")
    d_ptr = torch.full((shift_by + total + 8,), -1, dtype=torch.int64, device="cuda") if mode == 1 else None
    d_doc = torch.full((shift_by + total + 8,), -1, dtype=torch.int32, device="cuda") if doc else None
    d_cls = torch.zeros((60, 2), dtype=torch.int64, device="cuda") if mode == 0 else None
    ix.query_device(mode, d_seqs, d_offs, total, d_lengths=d_len if mode == 0 else None, d_pointers=d_ptr, d_docs=d_doc,
                    d_class=d_cls, bin_width=150, max_value_thr=5)
    torch.cuda.synchronize()
    assert ix.last_chunk_stats()["chunk_len"] == 128
    sl = slice(shift_by, shift_by + total)
    if mode == 0:
        want = orc.pml(hs, ho, want_docs=doc)
        lens, docs = want if doc else (want, None)
        assert np.array_equal(d_len[sl].cpu().numpy().view(np.uint32), lens)
        assert bool((d_len[:shift_by] == -1).all())  # nothing written in front of the first read
        if doc:
            assert np.array_equal(d_doc[sl].cpu().numpy().view(np.uint32), docs)
        f, a, b, ssum = oracle_mod.classify(lens, ho, 150, 5)
        cls = d_cls.cpu().numpy().view(capi.CLASS_DTYPE).reshape(-1)
        assert np.array_equal(cls["above"], a) and np.array_equal(cls["below"], b)
    else:
        want = orc.ms(hs, ho, want_docs=True)
        assert np.array_equal(d_ptr[sl].cpu().numpy().view(np.uint64), want["pointers"])
        assert np.array_equal(d_doc[sl].cpu().numpy().view(np.uint32), want["docs"])


@pytest.mark.parametrize("out_dtype", [torch.int16, torch.int32])
def test_classifier_of_chunked_batches_over_bin_widths(oracle_mod, out_dtype):
    """The classifier that follows the chunked walk (k_classify_tiles: tiles of 512 / 256 lengths, bins collected in LDS; bins
    narrower than 8 stay on the lane-per-bin kernel): bins narrower than, equal to, not dividing and wider than a tile, wider
    than a read, reads of ragged lengths (a last bin that takes the remainder, reads shorter than a bin, empty reads), against
    the oracle's classifier over the oracle's lengths."""
    raw = synth.statistical_rlbwt(1 << 16, 253, 8.0, seed=6, device="cuda", zipf=1.0)
    ix = _chunked(capi.Index.from_raw(raw, 0), 128)
    orc = oracle_mod.OracleIndex.from_raw(raw.cpu())
    seqs, offs = synth.simulate_reads(raw, 400, 2200, seed=5)
    # ragged: cut every read to a length of its own (0 .. 2200), keeping the characters where they are
    rng = np.random.default_rng(3)
    lens = rng.integers(0, 2201, size=400)
    lens[:6] = [0, 1, 7, 8, 511, 513]
    o_h = offs.cpu().numpy()
    s_h = seqs.cpu().numpy()
    parts = [s_h[o_h[i]: o_h[i] + lens[i]] for i in range(400)]
    s2 = np.concatenate(parts).astype(np.uint8)
    o2 = np.zeros(401, dtype=np.uint64)
    o2[1:] = np.cumsum(lens)
    want = orc.pml(s2, o2)
    d_seqs = capi.pad_seqs(torch.from_numpy(s2).cuda())
    d_offs = torch.from_numpy(o2.view(np.int64)).cuda()
    for w in (1, 5, 8, 9, 16, 37, 150, 256, 512, 513, 1000, 2199, 2200, 5000):
        for thr in (1, 5):
            d_len = torch.empty(s2.size + 8, dtype=out_dtype, device="cuda")
            d_cls = torch.empty((400, 2), dtype=torch.int64, device="cuda")
            ix.query_device(capi.SPX_MODE_PML, d_seqs, d_offs, int(s2.size), d_lengths=d_len, d_class=d_cls, bin_width=w, max_value_thr=thr)
            torch.cuda.synchronize()
            assert ix.last_chunk_stats()["chunk_len"] == 128
            f, a, b, s = oracle_mod.classify(want, o2, w, thr)
            c32 = d_cls.view(torch.int32).view(400, 4).cpu().numpy()
            assert np.array_equal(c32[:, 2].view(np.uint32), a) and np.array_equal(c32[:, 3].view(np.uint32), b), (w, thr)
            assert np.array_equal(d_cls[:, 0].cpu().numpy().view(np.uint64), s), (w, thr)
