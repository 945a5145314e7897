"""-m gpu: minimizer digestion on the device (spx_digest_*) against oracle/orc_digest.c, bit for
bit, and the digest -> query chain against the oracle's digest -> query."""
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


@pytest.fixture(scope="module")
def small_index(gpu):
    raw, text = cases.real_case(21, 4000, DNA)
    return raw, text, capi.Index.from_raw(raw, 0)


def _ragged_dna(rng, nreads, maxlen, p_n):
    lens = rng.integers(0, maxlen, size=nreads)
    lens[::13] = 0  # empty reads
    lens[1::29] = rng.integers(1, 4, size=lens[1::29].size)  # shorter than k
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    seqs = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(offs[-1]))].copy()
    seqs[rng.random(seqs.size) < p_n] = ord("N")
    return seqs, offs


@pytest.mark.parametrize("kind", [capi.SPX_DIGEST_PROMOTED, capi.SPX_DIGEST_DNA])
@pytest.mark.parametrize("k,w", [(4, 11), (3, 10), (2, 9), (1, 8), (4, 4), (1, 1), (2, 3), (3, 20), (4, 100), (1, 64), (4, 131)])
def test_digest_matches_oracle(gpu, oracle_mod, small_index, kind, k, w):
    ix = small_index[2]
    rng = np.random.default_rng(100 * k + w + kind)
    for p_n, maxlen in ((0.0, 700), (0.05, 700), (0.01, 120)):
        seqs, offs = _ragged_dna(rng, 400, maxlen, p_n)
        want, want_offs = oracle_mod.digest_batch(kind, k, w, seqs, offs)
        for forced in (0, 1, 2, 3):  # automatic, lane-per-read (when the window allows), wavefront-per-read, lane-per-chunk (k = 4, w = 11)
            ix.set_option("digest_kernel", forced)
            got, got_offs = ix.digest_host(kind, k, w, seqs, offs)
            assert np.array_equal(got_offs, want_offs), forced
            assert np.array_equal(got, want), forced
    ix.set_option("digest_kernel", 0)


def test_digest_edge_batches(gpu, oracle_mod, small_index):
    ix = small_index[2]
    # no reads at all / only empty reads / one read / one long read crossing many 64-character chunks
    for seqs, offs in (
        (np.zeros(0, np.uint8), np.array([0], np.uint64)),
        (np.zeros(0, np.uint8), np.array([0, 0, 0], np.uint64)),
        (np.frombuffer(b"ACGTACGTACGTAAAC", np.uint8), np.array([0, 16], np.uint64)),
        (np.frombuffer(b"N" * 70 + b"ACGTTGCAAGT" + b"N" * 130 + b"A" * 64 + b"C" * 64 + b"GATTACA" * 31, np.uint8),
         None),
    ):
        if offs is None:
            offs = np.array([0, seqs.size], np.uint64)
        for kind in (1, 2):
            want, want_offs = oracle_mod.digest_batch(kind, 4, 11, seqs, offs)
            for forced in (1, 2, 3, 0):
                ix.set_option("digest_kernel", forced)
                got, got_offs = ix.digest_host(kind, 4, 11, seqs, offs)
                assert np.array_equal(got_offs, want_offs) and np.array_equal(got, want)
    # lower-case and IUPAC letters are not ACGT: the caller upper-cases (compute_ms_pml.cpp:916-917)
    seqs = np.frombuffer(b"acgtacgtacgtacgtRYKMACGTACGTACGTACG", np.uint8)
    offs = np.array([0, seqs.size], np.uint64)
    want, want_offs = oracle_mod.digest_batch(1, 4, 11, seqs, offs)
    got, got_offs = ix.digest_host(1, 4, 11, seqs, offs)
    assert np.array_equal(got_offs, want_offs) and np.array_equal(got, want)


def test_charhash_option_and_argument_checks(gpu, oracle_mod, small_index):
    raw = small_index[0]
    ix = capi.Index.from_raw(raw, 0)
    rng = np.random.default_rng(3)
    seqs, offs = _ragged_dna(rng, 100, 300, 0.01)
    ch = [7, 201, 64, 130]
    ix.set_option("minimizer_charhash", ch[0] | ch[1] << 8 | ch[2] << 16 | ch[3] << 24)
    want, want_offs = oracle_mod.digest_batch(1, 4, 11, seqs, offs, charhash=ch)
    got, got_offs = ix.digest_host(1, 4, 11, seqs, offs)
    assert np.array_equal(got_offs, want_offs) and np.array_equal(got, want)
    # the default differs (the option took effect)
    d2, _ = capi.Index.from_raw(raw, 0).digest_host(1, 4, 11, seqs, offs)
    assert not np.array_equal(d2, got)
    for bad in ((0, 4, 11), (3, 4, 11), (1, 5, 11), (1, 0, 11), (1, 4, 3), (1, 4, 1 << 20)):
        with pytest.raises(capi.SpxError):
            ix.digest_host(bad[0], bad[1], bad[2], seqs, offs)


@pytest.mark.parametrize("kind", [capi.SPX_DIGEST_PROMOTED, capi.SPX_DIGEST_DNA])
def test_digest_then_query_equals_oracle_chain(gpu, oracle_mod, kind):
    """An index over the DIGESTED text, queried with raw DNA reads through digest + walk: the
    reference's per-read loop body (compute_ms_pml.cpp:916-938), batch form, host + device + fused."""
    rng = np.random.default_rng(40 + kind)
    genome = cases.repetitive_text(rng, 30000, DNA)
    k, w = 4, 11
    dtext = oracle_mod.digest(kind, k, w, genome)
    raw = synth.index_from_text(torch.from_numpy(dtext.copy()), doc_lengths=[dtext.size // 2, dtext.size - dtext.size // 2])
    orc = oracle_mod.OracleIndex.from_raw(raw)
    ix = capi.Index.from_raw(raw, 0)
    seqs, offs = cases.reads_mixed(rng, genome, DNA, 500, 400, [ord("N")])
    offs = offs.astype(np.uint64)
    dseqs, doffs = oracle_mod.digest_batch(kind, k, w, seqs, offs)
    want_l, want_d = orc.pml(dseqs, doffs.astype(np.int64), want_docs=True)
    want_ms = orc.ms(dseqs, doffs.astype(np.int64), want_docs=True, text=dtext)
    # fused host call
    got = ix.digest_query_host(capi.SPX_MODE_PML, kind, k, w, seqs, offs, want_docs=True, classify=(5, 2))
    assert np.array_equal(got["offsets"], doffs)
    assert np.array_equal(got["lengths"], want_l)
    assert np.array_equal(got["docs"], want_d)
    f, a, b, s = oracle_mod.classify(want_l, doffs.astype(np.int64), 5, 2)
    assert np.array_equal(got["class"]["above"], a) and np.array_equal(got["class"]["sum_max"], s)
    ix.set_text(torch.from_numpy(dtext.copy()))
    got = ix.digest_query_host(capi.SPX_MODE_MS, kind, k, w, seqs, offs, want_docs=True)
    assert np.array_equal(got["pointers"], want_ms["pointers"])
    assert np.array_equal(got["lengths"], want_ms["lengths"])
    assert np.array_equal(got["docs"], want_ms["docs"])
    # device chain: nothing visits the host between digestion and the walk
    d_seqs = torch.from_numpy(seqs).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    d_dseqs, d_doffs = ix.digest_device(kind, k, w, d_seqs, d_offs, int(offs[-1]))
    total = int(d_doffs[-1].item())
    assert total == int(doffs[-1])
    d_len = torch.zeros(total + 8, dtype=torch.int32, device="cuda")
    ix.query_device(capi.SPX_MODE_PML, d_dseqs, d_doffs, total, d_lengths=d_len)
    torch.cuda.synchronize()
    assert np.array_equal(d_len[:total].cpu().numpy().astype(np.uint32), want_l)
    assert np.array_equal(d_dseqs[:total].cpu().numpy(), dseqs)


@pytest.mark.parametrize("kind", [capi.SPX_DIGEST_PROMOTED, capi.SPX_DIGEST_DNA])
def test_digest_query_on_the_device_parked_and_concatenated(gpu, oracle_mod, kind):
    """spx_digest_query_batch_device[16]: DNA reads resident on the device -> digestion -> walk in one call.  With
    "digest_parked" = 2 the digested reads stay where the digestion parked them and the walk takes them by the reads' input
    offsets (no concatenation pass; -m only), with 1 they are concatenated first: the same offsets, PML values, document ids,
    classes, MS pointers -- the oracle's (compute_ms_pml.cpp:919-938, batch form).  Ragged reads, empty ones, reads shorter
    than a window, characters outside ACGT."""
    rng = np.random.default_rng(60 + kind)
    genome = cases.repetitive_text(rng, 30000, DNA)
    k, w = 4, 11
    dtext = oracle_mod.digest(kind, k, w, genome)
    raw = synth.index_from_text(torch.from_numpy(dtext.copy()), doc_lengths=[dtext.size // 2, dtext.size - dtext.size // 2])
    orc = oracle_mod.OracleIndex.from_raw(raw)
    ix = capi.Index.from_raw(raw, 0)
    seqs, offs = cases.reads_mixed(rng, genome, DNA, 3000, 300, [ord("N")])
    offs = offs.astype(np.uint64)
    dseqs, doffs = oracle_mod.digest_batch(kind, k, w, seqs, offs)
    want_l, want_d = orc.pml(dseqs, doffs.astype(np.int64), want_docs=True)
    want_ms = orc.ms(dseqs, doffs.astype(np.int64), want_docs=True)
    f, a, b, sm = oracle_mod.classify(want_l, doffs.astype(np.int64), 5, 2)
    total_in = int(offs[-1])
    d_seqs = torch.zeros(total_in + 64, dtype=torch.uint8, device="cuda")
    d_seqs[:total_in] = torch.from_numpy(seqs).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    nreads = offs.size - 1
    for parked in (2, 1, 0):
        ix.set_option("digest_parked", parked)
        for vt in (torch.int16, torch.int32):
            d_len = torch.full((total_in + 8,), -1, dtype=vt, device="cuda")
            d_doc = torch.full((total_in + 8,), -1, dtype=vt, device="cuda")
            d_cls = torch.zeros((nreads, 2), dtype=torch.int64, device="cuda")
            d_oo, _ = ix.digest_query_device(capi.SPX_MODE_PML, kind, k, w, d_seqs, d_offs, total_in, d_lengths=d_len, d_docs=d_doc,
                                             d_class=d_cls, bin_width=5, max_value_thr=2)
            torch.cuda.synchronize()
            assert np.array_equal(d_oo.cpu().numpy().astype(np.uint64), doffs), (parked, vt)
            tot = int(doffs[-1])
            mask = 0xffff if vt == torch.int16 else 0xffffffff
            assert np.array_equal(d_len[:tot].cpu().numpy().astype(np.int64) & mask, want_l.astype(np.int64)), (parked, vt)
            assert np.array_equal(d_doc[:tot].cpu().numpy().astype(np.int64) & mask, want_d.astype(np.int64)), (parked, vt)
            cls = d_cls.cpu().numpy().view(capi.CLASS_DTYPE).reshape(-1)
            assert np.array_equal(cls["above"], a) and np.array_equal(cls["sum_max"], sm), (parked, vt)
        # MS pointers + document ids without lengths: the walk alone, so the parked reads serve it too
        d_ptr = torch.zeros(total_in + 8, dtype=torch.int64, device="cuda")
        d_doc = torch.zeros(total_in + 8, dtype=torch.int32, device="cuda")
        ix.digest_query_device(capi.SPX_MODE_MS, kind, k, w, d_seqs, d_offs, total_in, d_pointers=d_ptr, d_docs=d_doc)
        torch.cuda.synchronize()
        tot = int(doffs[-1])
        assert np.array_equal(d_ptr[:tot].cpu().numpy().view(np.uint64), want_ms["pointers"]), parked
        assert np.array_equal(d_doc[:tot].cpu().numpy().view(np.uint32), want_ms["docs"]), parked
    ix.set_option("digest_parked", 0)


@pytest.mark.parametrize("kind", [capi.SPX_DIGEST_PROMOTED, capi.SPX_DIGEST_DNA])
def test_long_reads_are_digested_by_chunks(gpu, oracle_mod, small_index, kind):
    """Reads of thousands of characters (BASELINE config 5 before digestion) take the lane-per-chunk kernel: every chunk of 240
    characters a lane, started twelve characters early, its bytes parked, one scan, the pieces moved -- the same bytes and
    offsets as the oracle's sequential loop (src/spumoni.cpp:294-342), also for reads whose length is a multiple of the chunk,
    one character more or less, shorter than a window, empty, and for reads with characters outside ACGT (flagged, redone by
    the wavefront-per-read kernel)."""
    ix = small_index[2]
    rng = np.random.default_rng(300 + kind)
    lens = [9000, 240, 241, 239, 480, 0, 7, 11, 10, 12, 2400, 2399, 2401, 5000, 1, 252, 15000, 3333]
    lens += rng.integers(2500, 12000, size=40).tolist()
    reads = [np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=n)].copy() for n in lens]
    for i in (3, 13, 20, 21, 35):  # characters outside ACGT: at a chunk boundary, in a halo, in the middle, at the end
        if reads[i].size:
            for at in (0, 239, 240, 251, reads[i].size // 2, reads[i].size - 1):
                if at < reads[i].size:
                    reads[i][at] = ord("N")
    seqs = np.concatenate(reads)
    offs = np.concatenate([[0], np.cumsum([r.size for r in reads])]).astype(np.uint64)
    want, want_offs = oracle_mod.digest_batch(kind, 4, 11, seqs, offs)
    for forced in (0, 3, 2):
        ix.set_option("digest_kernel", forced)
        got, got_offs = ix.digest_host(kind, 4, 11, seqs, offs)
        assert np.array_equal(got_offs, want_offs), forced
        assert np.array_equal(got, want), forced
    ix.set_option("digest_kernel", 0)


def test_digest_large_batch_properties(gpu, oracle_mod, small_index):
    """2*10^6 reads x 200 bp: sizes the oracle does not visit -- sampled reads against the oracle,
    plus size-independent properties (idempotent offsets, alphabet, density)."""
    ix = small_index[2]
    g = torch.Generator(device="cuda").manual_seed(7)
    nreads, m = 2_000_000, 200
    d_seqs = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")[
        torch.randint(0, 4, (nreads * m + 64,), generator=g, device="cuda")]
    d_offs = torch.arange(nreads + 1, dtype=torch.int64, device="cuda") * m
    out, out_offs = ix.digest_device(1, 4, 11, d_seqs, d_offs, nreads * m)
    out2, out_offs2 = ix.digest_device(1, 4, 11, d_seqs, d_offs, nreads * m)
    torch.cuda.synchronize()
    assert torch.equal(out_offs, out_offs2)
    total = int(out_offs[-1].item())
    assert torch.equal(out[:total], out2[:total])
    assert int(out[:total].min().item()) >= 3
    lens = (out_offs[1:] - out_offs[:-1]).double()
    assert 0.18 * m < lens.mean().item() < 0.26 * m  # about 2 / (wsz + 1) per base
    assert int(out[total:total + 32].max().item()) == 0  # read-ahead tail is defined
    pick = torch.randint(0, nreads, (300,), generator=torch.Generator().manual_seed(1)).tolist()
    h_seqs = d_seqs.cpu().numpy()
    h_out = out[:total].cpu().numpy()
    h_offs = out_offs.cpu().numpy()
    for q in pick:
        want = oracle_mod.digest(1, 4, 11, h_seqs[q * m:(q + 1) * m])
        assert np.array_equal(h_out[h_offs[q]:h_offs[q + 1]], want)
