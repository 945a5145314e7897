"""-m gpu: BASELINE configs at (or near) their full sizes.  The oracle checks a sample bit for
bit; the whole batch is checked through size-independent properties of the walk:
 * partition invariance  -- a read's result does not depend on which batch / lane it ran in;
 * suffix property       -- the walk starts at the right end of a read, so PML(read[j:]) ==
                            PML(read)[j:] for every j (same for MS pointers);
 * classification        -- reads sampled from the text are FOUND, reversed ones are not.
"""
import numpy as np
import pytest
import torch

from spumoni_amd import capi, synth
from tests import cases

pytestmark = pytest.mark.gpu


def _pml_dev(ix, seqs_t, offs_t, classify=None):
    d_seqs = capi.pad_seqs(seqs_t)
    n = seqs_t.numel()
    d_len = torch.empty(max(n, 1), dtype=torch.int32, device="cuda")
    d_cls = torch.empty((offs_t.numel() - 1, 2), dtype=torch.int64, device="cuda") if classify else None
    bw, thr = classify if classify else (0, 0)
    ix.query_device(capi.SPX_MODE_PML, d_seqs, offs_t, n, d_lengths=d_len, d_class=d_cls, bin_width=bw, max_value_thr=thr)
    torch.cuda.synchronize()
    ix.last_stats()  # raises if the walk flagged an inconsistency
    return d_len[:n], d_cls


def test_config2_five_strain_ecoli_1m_reads(oracle_mod):
    """BASELINE config[1]: 5-strain E. coli pangenome (+revcomp, 46.4 Mbp), 1M x 200 bp, PML."""
    base = synth.random_genome(4_641_652, seed=1)
    genomes = [base] + [synth.mutate(base, seed=s) for s in (2, 3, 4, 5)]
    text, doc_lengths = synth.pangenome_text(genomes)
    raw = synth.index_from_text(torch.from_numpy(text).cuda(), doc_lengths=doc_lengths, with_samples=False)
    assert raw.n == text.size + 1
    nreads, m = 1_000_000, 200
    seqs, offs = synth.sample_reads(text, nreads, m, seed=12)
    ix = capi.Index.from_raw(raw, 0)
    d_seqs, d_offs = torch.from_numpy(seqs).cuda(), torch.from_numpy(offs).cuda()
    full, cls = _pml_dev(ix, d_seqs, d_offs, classify=(150, 7))
    # (1) oracle on a sample, bit-exact
    ns = 20_000
    orc = oracle_mod.OracleIndex.from_raw(raw.cpu())
    want = orc.pml(seqs[: ns * m], offs[: ns + 1])
    assert np.array_equal(full[: ns * m].cpu().numpy().view(np.uint32), want)
    # (2) partition invariance: second half of the reads as its own batch
    h = nreads // 2
    half, _ = _pml_dev(ix, d_seqs[h * m :], d_offs[h:] - d_offs[h])
    assert torch.equal(half, full[h * m :])
    # (3) suffix property on 100k reads cut at random positions
    rng = np.random.default_rng(0)
    cut = torch.from_numpy(rng.integers(1, m, size=100_000)).cuda()
    idx = torch.arange(100_000, device="cuda")
    lens = m - cut
    so = torch.zeros(100_001, dtype=torch.int64, device="cuda")
    so[1:] = torch.cumsum(lens, 0)
    pos = torch.arange(int(so[-1]), device="cuda")
    rid = torch.searchsorted(so, pos, right=True) - 1
    src = rid * m + cut[rid] + (pos - so[rid])
    sfx, _ = _pml_dev(ix, d_seqs[src], so)
    assert torch.equal(sfx, full[src])
    # (4) classification separates sampled (FOUND) from reversed (NOT_PRESENT) reads
    c32 = cls.view(torch.int32).view(-1, 4)
    found = (2 * c32[:, 2] > c32[:, 2] + c32[:, 3]).float().mean().item()
    assert 0.45 < found < 0.55
    st = ix.last_stats()
    assert st["steps"] == lens.sum().item()


def test_config3_shape_properties(oracle_mod):
    """BASELINE config[2] shape at r = 2^25 (the bench itself gates r = 2^28 against the oracle)."""
    raw = synth.statistical_rlbwt(1 << 25, 253, 8.0, seed=3, device="cuda", zipf=1.0)
    seqs, offs = synth.simulate_reads(raw, 2_000_000, 44, seed=13)
    ix = capi.Index.from_raw(raw, 0)
    full, _ = _pml_dev(ix, seqs, offs)
    ns = 50_000
    orc = oracle_mod.OracleIndex.from_raw(raw.cpu())
    want = orc.pml(seqs[: ns * 44].cpu().numpy(), offs[: ns + 1].cpu().numpy())
    assert np.array_equal(full[: ns * 44].cpu().numpy().view(np.uint32), want)
    # partition invariance with ragged split
    k = 777_777
    a, _ = _pml_dev(ix, seqs[: k * 44], offs[: k + 1])
    b, _ = _pml_dev(ix, seqs[k * 44 :], offs[k:] - offs[k])
    assert torch.equal(torch.cat([a, b]), full)


def test_config3_declared_size(oracle_mod):
    """BASELINE config[2] AT ITS DECLARED SIZE: statistical RLBWT r = 10^9, sigma = 253, Zipf(1.0) heads, mean run 8,
    seed 3; 10^7 pre-digested reads x 44 minimizer characters, seed 13, half simulated-positive after 4 warm-up characters --
    the index and the batch of bench.py's headline, through the entry point it times (spx_query_batch_device16 + classes).
    Oracle on the first 50 000 reads, bit for bit (lengths and the report's columns); partition invariance (ragged split) on all
    10^7.  What compute_ms_pml.cpp:238-286 computes; needs the whole device (230 GB of index) and ~40 GB of host memory."""
    import gc

    gc.collect()
    torch.cuda.empty_cache()
    free, total_mem = torch.cuda.mem_get_info()
    if total_mem < 250e9 or free < 0.9 * total_mem:
        pytest.skip("the declared C3 index needs a whole 288 GB device")
    r, m, nreads = 1_000_000_000, 44, 10_000_000
    raw = synth.statistical_rlbwt(r, 253, 8.0, seed=3, device="cuda", zipf=1.0)
    seqs, offs = synth.simulate_reads(raw, nreads, m, seed=13, positive_fraction=0.5, f_mis=0.02, warmup=4)
    rawc = raw.cpu()
    torch.cuda.empty_cache()
    ix = capi.Index.from_raw(raw, 0)
    del raw
    torch.cuda.empty_cache()
    assert ix.describe()["flat_runs"] >= r

    def pml16(s, o):
        n = s.numel()
        d_len = torch.empty(n + 8, dtype=torch.int16, device="cuda")
        d_cls = torch.empty((o.numel() - 1, 2), dtype=torch.int64, device="cuda")
        ix.query_device(capi.SPX_MODE_PML, capi.pad_seqs(s), o, n, d_lengths=d_len, d_class=d_cls, bin_width=150, max_value_thr=5)
        torch.cuda.synchronize()
        ix.last_stats()
        return d_len[:n], d_cls

    full, cls = pml16(seqs, offs)
    st = ix.last_stats()
    assert st["steps"] == nreads * m
    ns = 50_000
    orc = oracle_mod.OracleIndex.from_raw(rawc)
    want = orc.pml(seqs[: ns * m].cpu().numpy(), offs[: ns + 1].cpu().numpy())
    assert np.array_equal(full[: ns * m].cpu().numpy().view(np.uint16).astype(np.uint32), want)
    f, ab, be, sm = oracle_mod.classify(want, offs[: ns + 1].cpu().numpy(), 150, 5)
    c32 = cls[:ns].cpu().numpy().view(capi.CLASS_DTYPE).reshape(-1)
    assert np.array_equal(c32["above"], ab) and np.array_equal(c32["below"], be) and np.array_equal(c32["sum_max"], sm)
    del orc, rawc
    k = 3_777_777
    a, ca = pml16(seqs[: k * m], offs[: k + 1])
    b, cb = pml16(seqs[k * m:], offs[k:] - offs[k])
    assert torch.equal(torch.cat([a, b]), full) and torch.equal(torch.cat([ca, cb]), cls)
    ix.close()


def test_config5_long_reads(oracle_mod):
    """BASELINE config[4] shape: few long reads (latency-bound small-batch launch geometry)."""
    raw = synth.statistical_rlbwt(1 << 22, 253, 8.0, seed=6, device="cuda", zipf=1.0)
    seqs, offs = synth.simulate_reads(raw, 700, 2200, seed=16)
    ix = capi.Index.from_raw(raw, 0)
    full, _ = _pml_dev(ix, seqs, offs, classify=(150, 5))
    orc = oracle_mod.OracleIndex.from_raw(raw.cpu())
    want = orc.pml(seqs.cpu().numpy(), offs.cpu().numpy())
    assert np.array_equal(full.cpu().numpy().view(np.uint32), want)
    # a read longer than 65535 characters takes the unstaged output path
    long_seqs, long_offs = synth.simulate_reads(raw, 3, 70_000, seed=17, positive_fraction=0.0)
    h = raw.heads[torch.randint(1, raw.r, (40_000,), device="cuda")]  # make part of one read match-rich
    long_seqs[100:40_100] = h
    got, _ = _pml_dev(ix, long_seqs, long_offs)
    want = orc.pml(long_seqs.cpu().numpy(), long_offs.cpu().numpy())
    assert np.array_equal(got.cpu().numpy().view(np.uint32), want)


def test_host_entry_points_pipeline_large_batches(oracle_mod):
    """spx_query_batch / spx_query_batch16 run batches of >= 2^18 reads and >= 64 MB as a pipeline of
    chunks over three streams: same results as the device entry point (PML + doc + classes, MS
    pointers / lengths), and as the oracle on a sample."""
    rng = np.random.default_rng(4)
    raw, text = cases.real_case(9, 200_000, list(b"ACGT"), ndocs=5)
    ix = capi.Index.from_raw(raw, 0)
    nreads, m = 640_000, 130  # ragged below
    lens = rng.integers(90, m + 1, size=nreads)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    starts = rng.integers(0, text.size - m, size=nreads)
    seqs = np.empty(int(offs[-1]), dtype=np.uint8)
    big = text[(starts[:, None] + np.arange(m)[None, :])]
    mask = np.arange(m)[None, :] < lens[:, None]
    seqs[:] = big[mask]
    flip = rng.random(seqs.size) < 0.02
    seqs[flip] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(flip.sum()))]
    assert seqs.size >= (64 << 20)
    d_seqs, d_offs = torch.from_numpy(seqs).cuda(), torch.from_numpy(offs).cuda()
    want_l, want_c = _pml_dev(ix, d_seqs, d_offs, classify=(50, 4))
    want_l = want_l.cpu().numpy().view(np.uint32)
    cls32 = want_c.view(torch.int32).view(nreads, 4).cpu().numpy()
    for bits in (32, 16):
        got = ix.query_host(capi.SPX_MODE_PML, seqs, offs, want_docs=True, classify=(50, 4), bits=bits)
        assert np.array_equal(got["lengths"], want_l)
        assert np.array_equal(got["class"]["above"], cls32[:, 2].view(np.uint32))
        assert np.array_equal(got["class"]["below"], cls32[:, 3].view(np.uint32))
        ms = ix.query_host(capi.SPX_MODE_MS, seqs, offs, want_docs=True, bits=bits)
        if bits == 32:
            ref_ms, ref_docs = ms, got["docs"]
        else:
            assert np.array_equal(ms["pointers"], ref_ms["pointers"])
            assert np.array_equal(ms["lengths"], ref_ms["lengths"])
            assert np.array_equal(ms["docs"], ref_ms["docs"]) and np.array_equal(got["docs"], ref_docs)
    ns = 3000
    orc = oracle_mod.OracleIndex.from_raw(raw)
    wl, wd = orc.pml(seqs[: offs[ns]], offs[: ns + 1], want_docs=True)
    assert np.array_equal(want_l[: offs[ns]], wl) and np.array_equal(ref_docs[: offs[ns]], wd)
    w = orc.ms(seqs[: offs[ns]], offs[: ns + 1], want_docs=True, text=text)
    assert np.array_equal(ref_ms["pointers"][: offs[ns]], w["pointers"])
    assert np.array_equal(ref_ms["lengths"][: offs[ns]], w["lengths"])
    assert np.array_equal(ref_ms["docs"][: offs[ns]], w["docs"])


def test_pipelined_16_bit_batch_with_a_read_too_long_is_refused():
    """The 16-bit host entry point on a pipelined batch (>= 2^18 reads, >= 64 MB): the reads' lengths are checked beside
    the pipeline, not in front of it -- a read of 65536 characters or more still fails the call with the same error
    (and the 32-bit entry point takes the batch)."""
    raw, text = cases.real_case(19, 50_000, list(b"ACGT"))
    ix = capi.Index.from_raw(raw, 0)
    rng = np.random.default_rng(6)
    nreads = (1 << 18) + 3
    lens = np.full(nreads, 260, dtype=np.int64)
    lens[nreads - 7] = 70_000
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    seqs = text[rng.integers(0, text.size, size=int(offs[-1]))]
    assert seqs.size >= (64 << 20)
    with pytest.raises(capi.SpxError, match=f"read {nreads - 7} has 65536 characters or more"):
        ix.query_host(capi.SPX_MODE_PML, seqs, offs, bits=16)
    got = ix.query_host(capi.SPX_MODE_PML, seqs, offs, bits=32)
    lens[nreads - 7] = 260  # without the long read the 16-bit call goes through, same values for the reads before it
    offs2 = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    cut = int(offs[nreads - 7])
    seqs2 = np.concatenate([seqs[:cut], seqs[cut: cut + 260], seqs[int(offs[nreads - 6]):]])
    got16 = ix.query_host(capi.SPX_MODE_PML, seqs2, offs2, bits=16)
    assert np.array_equal(got16["lengths"][:cut], got["lengths"][:cut])


def test_config4_scale_ms_doc(oracle_mod):
    """BASELINE config[3] shape at scale: statistical index r = 2^27 with SA samples and 10 documents, 5 * 10^6
    reads of 55 minimizer characters, MS pointers + document ids (MS lengths need a text, which a statistical
    index does not have: test_config2 / the parity tests cover them).  Oracle on a 50 000-read sample, bit for
    bit; the whole batch through partition invariance and the suffix property (the walk starts at a read's
    right end, so pointers / documents of a read's suffix are the suffix of the read's)."""
    raw = synth.statistical_rlbwt(1 << 27, 253, 8.0, seed=5, device="cuda", zipf=1.0, with_samples=True, n_docs=10)
    nreads, m = 5_000_000, 55
    seqs, offs = synth.simulate_reads(raw, nreads, m, seed=15, warmup=4)
    ix = capi.Index.from_raw(raw, 0)
    d_seqs = capi.pad_seqs(seqs)

    def ms_dev(s, o):
        n = s.numel()
        d_ptr = torch.empty(n, dtype=torch.int64, device="cuda")
        d_doc = torch.empty(n + 8, dtype=torch.int32, device="cuda")
        ix.query_device(capi.SPX_MODE_MS, capi.pad_seqs(s), o, n, d_pointers=d_ptr, d_docs=d_doc)
        torch.cuda.synchronize()
        ix.last_stats()
        return d_ptr, d_doc[:n]

    ptr, doc = ms_dev(seqs, offs)
    ns = 50_000
    rawc = raw.cpu()
    del raw
    orc = oracle_mod.OracleIndex.from_raw(rawc)
    w = orc.ms(seqs[: ns * m].cpu().numpy(), offs[: ns + 1].cpu().numpy(), want_docs=True)
    assert np.array_equal(ptr[: ns * m].cpu().numpy().view(np.uint64), w["pointers"])
    assert np.array_equal(doc[: ns * m].cpu().numpy().view(np.uint32), w["docs"])
    # partition invariance, ragged split
    k = 1_777_777
    pa, da = ms_dev(seqs[: k * m], offs[: k + 1])
    pb, db = ms_dev(seqs[k * m:], offs[k:] - offs[k])
    assert torch.equal(torch.cat([pa, pb]), ptr) and torch.equal(torch.cat([da, db]), doc)
    # suffix property on 200k reads cut at random positions
    rng = np.random.default_rng(1)
    nc = 200_000
    cut = torch.from_numpy(rng.integers(1, m, size=nc)).cuda()
    lens = m - cut
    so = torch.zeros(nc + 1, dtype=torch.int64, device="cuda")
    so[1:] = torch.cumsum(lens, 0)
    pos = torch.arange(int(so[-1]), device="cuda")
    rid = torch.searchsorted(so, pos, right=True) - 1
    src = rid * m + cut[rid] + (pos - so[rid])
    ps, ds = ms_dev(seqs[src], so)
    assert torch.equal(ps, ptr[src]) and torch.equal(ds, doc[src])


def test_config4_declared_size(oracle_mod):
    """BASELINE config[3] AT ITS DECLARED SIZE: the C3 index (r = 10^9, seed 3) + SA samples + 10 documents, 5 * 10^6 reads
    x 55 minimizer characters (250 bp), MS pointers (u64) + document ids (u16) -- bench.py's c4_ms_doc leg as a test.  Oracle on the
    first 20 000 reads, bit for bit (compute_ms_pml.cpp:626-682); partition invariance (ragged split) on all 5 * 10^6.  MS LENGTHS need a
    text, which a statistical index does not have: test_config2 / the parity tests / real_bwt_ms_doc cover them."""
    import gc

    gc.collect()
    torch.cuda.empty_cache()
    free, total_mem = torch.cuda.mem_get_info()
    if total_mem < 250e9 or free < 0.9 * total_mem:
        pytest.skip("the declared C4 index needs a whole 288 GB device")
    r, m, nreads = 1_000_000_000, 55, 5_000_000
    raw = synth.statistical_rlbwt(r, 253, 8.0, seed=3, device="cuda", zipf=1.0, with_samples=True, n_docs=10)
    seqs, offs = synth.simulate_reads(raw, nreads, m, seed=14, positive_fraction=0.5, f_mis=0.02, warmup=4)
    rawc = raw.cpu()
    torch.cuda.empty_cache()
    ix = capi.Index.from_raw(raw, 0)
    del raw
    torch.cuda.empty_cache()
    d = ix.describe()
    assert d["has_samples"] and d["has_docs"] and d["fat_stride"] == 32

    def ms16(s, o):
        n = s.numel()
        d_ptr = torch.empty(n, dtype=torch.int64, device="cuda")
        d_doc = torch.empty(n + 8, dtype=torch.int16, device="cuda")
        ix.query_device(capi.SPX_MODE_MS, capi.pad_seqs(s), o, n, d_pointers=d_ptr, d_docs=d_doc)
        torch.cuda.synchronize()
        ix.last_stats()
        return d_ptr, d_doc[:n]

    ptr, doc = ms16(seqs, offs)
    ns = 20_000
    orc = oracle_mod.OracleIndex.from_raw(rawc)
    w = orc.ms(seqs[: ns * m].cpu().numpy(), offs[: ns + 1].cpu().numpy(), want_docs=True)
    assert np.array_equal(ptr[: ns * m].cpu().numpy().view(np.uint64), w["pointers"])
    assert np.array_equal(doc[: ns * m].cpu().numpy().view(np.uint16).astype(np.uint32), w["docs"])
    del orc, rawc
    k = 1_777_777
    pa, da = ms16(seqs[: k * m], offs[: k + 1])
    pb, db = ms16(seqs[k * m:], offs[k:] - offs[k])
    assert torch.equal(torch.cat([pa, pb]), ptr) and torch.equal(torch.cat([da, db]), doc)
    ix.close()


def test_config5_scale_long_reads_chunked(oracle_mod):
    """BASELINE config[4] shape at scale: 50 000 reads of 2 200 minimizer characters (10 kbp at the digestion
    density) and the per-GPU share of 6 250, statistical index r = 2^27.  Such batches take the chunked walk
     on their own; it must equal the plain walk of the same batch bit for bit (the whole batch:
    lengths and classes), and the oracle on a sample of whole reads."""
    raw = synth.statistical_rlbwt(1 << 27, 253, 8.0, seed=6, device="cuda", zipf=1.0)
    ix = capi.Index.from_raw(raw, 0)
    rawc = raw.cpu()
    del raw
    orc = oracle_mod.OracleIndex.from_raw(rawc)
    for nreads in (50_000, 6_250):
        seqs, offs = synth.simulate_reads(_dev_raw(rawc), nreads, 2200, seed=16 + nreads, warmup=4)
        ix.set_option("chunk_mode", 0)
        got, cls = _pml_dev(ix, seqs, offs, classify=(150, 5))
        cs = ix.last_chunk_stats()
        assert cs["chunk_len"] >= 128, "a long-read batch must take the chunked walk"
        assert cs["fallback_reads"] <= nreads // 1000
        ix.set_option("chunk_mode", 1)  # never: the plain walk
        want, wcls = _pml_dev(ix, seqs, offs, classify=(150, 5))
        assert ix.last_chunk_stats()["chunk_len"] == 0
        assert torch.equal(got, want) and torch.equal(cls, wcls)
        ns = 150
        o = orc.pml(seqs[: ns * 2200].cpu().numpy(), offs[: ns + 1].cpu().numpy())
        assert np.array_equal(got[: ns * 2200].cpu().numpy().view(np.uint32), o)
    ix.set_option("chunk_mode", 0)


def test_config5_declared_size(oracle_mod):
    """BASELINE config[4] AT ITS DECLARED SIZE: statistical RLBWT r = 2 * 10^9 (the 20-haplotype index), seed 6; 50 000
    reads x 2 200 minimizer characters (10 kbp at the digestion density), seed 17, and the per-GPU share of 6 250 -- bench.py's
    long_reads_c5 leg as a test.  The chunked walk (what such batches take by themselves) against the plain walk of the same batch,
    bit for bit over the whole batch (lengths and classes); the oracle on the first 400 reads (compute_ms_pml.cpp:238-286)."""
    import gc

    gc.collect()
    torch.cuda.empty_cache()
    free, total_mem = torch.cuda.mem_get_info()
    if total_mem < 250e9 or free < 0.9 * total_mem:
        pytest.skip("the declared C5 index needs a whole 288 GB device")
    r = 2_000_000_000
    raw = synth.statistical_rlbwt(r, 253, 8.0, seed=6, device="cuda", zipf=1.0)
    seqs, offs = synth.simulate_reads(raw, 56_250, 2200, seed=17, positive_fraction=0.5, f_mis=0.02, warmup=4)
    rawc = raw.cpu()
    torch.cuda.empty_cache()
    ix = capi.Index.from_raw(raw, 0)
    del raw
    torch.cuda.empty_cache()
    assert ix.describe()["flat_runs"] >= r
    orc = oracle_mod.OracleIndex.from_raw(rawc)
    del rawc
    for lo_r, hi_r in ((0, 6_250), (6_250, 56_250)):
        s = seqs[lo_r * 2200: hi_r * 2200].contiguous()
        o = (offs[lo_r: hi_r + 1] - offs[lo_r]).contiguous()
        ix.set_option("chunk_mode", 0)
        got, cls = _pml_dev(ix, s, o, classify=(150, 5))
        cs = ix.last_chunk_stats()
        assert cs["chunk_len"] >= 128, "a long-read batch must take the chunked walk"
        assert cs["fallback_reads"] <= (hi_r - lo_r) // 1000
        ix.set_option("chunk_mode", 1)
        want, wcls = _pml_dev(ix, s, o, classify=(150, 5))
        assert torch.equal(got, want) and torch.equal(cls, wcls)
        ns = 200
        w = orc.pml(s[: ns * 2200].cpu().numpy(), o[: ns + 1].cpu().numpy())
        assert np.array_equal(got[: ns * 2200].cpu().numpy().view(np.uint32), w)
    ix.set_option("chunk_mode", 0)
    ix.close()


def _dev_raw(rawc):
    """the raw arrays back on the device (simulate_reads runs where they live)"""
    kw = {}
    import dataclasses

    for f in dataclasses.fields(rawc):
        v = getattr(rawc, f.name)
        kw[f.name] = v.cuda() if isinstance(v, torch.Tensor) else v
    return synth.RawIndex(**kw)
