"""-m gpu: the output files' text written on the device (spx_query_text_begin / _fetch, spx_text.hip) against the
oracle's vectors formatted the reference's way: per read ">id\\n" then "<value> " per character and "\\n"
(/root/reference/src/compute_ms_pml.cpp:1001-1010, 1182-1205).  The ids never travel: the device leaves a gap."""
import numpy as np
import pytest

from spumoni_amd import capi, synth
from tests import cases

pytestmark = pytest.mark.gpu

DNA = list(b"ACGT")


def _expect(values, offs, ids):
    out = bytearray()
    for q, name in enumerate(ids):
        out += b">" + name + b"\n"
        out += b"".join(b"%d " % int(v) for v in values[offs[q]: offs[q + 1]]) + b"\n"
    return bytes(out)


def _fill(text, line_start, ids):
    b = bytearray(text)
    for q, name in enumerate(ids):
        at = int(line_start[q])
        b[at: at + len(name) + 2] = b">" + name + b"\n"
    return bytes(b)


@pytest.mark.parametrize("seed,letters", [(51, DNA + [ord("N")]), (52, [3, 4, 5, 90, 127, 128, 129, 200, 255])])
def test_text_streams_are_the_reference_bytes(oracle_mod, seed, letters):
    raw, text = cases.real_case(seed, 9000, letters, ndocs=5)
    rng = np.random.default_rng(seed)
    seqs, offs = cases.reads_mixed(rng, text, letters, 300, 500, [2])  # ragged, some empty
    ids = [b"read_%d%s" % (q, b" descr" * (q % 3)) for q in range(offs.size - 1)]
    gap = np.array([len(i) + 2 for i in ids], dtype=np.uint32)
    orc = oracle_mod.OracleIndex.from_raw(raw)
    ix = capi.Index.from_raw(raw, 0)
    # PML + document ids
    lens, docs = orc.pml(seqs, offs, want_docs=True)
    got = ix.query_text(capi.SPX_MODE_PML, seqs, offs, gap, capi.SPX_TEXT_LENGTHS | capi.SPX_TEXT_DOCS, classify=(20, 6))
    assert got["text"][1] is None
    assert _fill(got["text"][0], got["line_start"][0], ids) == _expect(lens, offs, ids)
    assert _fill(got["text"][2], got["line_start"][2], ids) == _expect(docs, offs, ids)
    f, a, b, s = oracle_mod.classify(lens, offs, 20, 6)
    assert np.array_equal(got["class"]["above"], a) and np.array_equal(got["class"]["below"], b)
    # MS: lengths, pointers (up to 13 digits), document ids
    ix.set_text(__import__("torch").from_numpy(text.copy()))
    w = orc.ms(seqs, offs, want_docs=True, text=text)
    got = ix.query_text(capi.SPX_MODE_MS, seqs, offs, gap, capi.SPX_TEXT_LENGTHS | capi.SPX_TEXT_POINTERS | capi.SPX_TEXT_DOCS)
    for i, key in enumerate(("lengths", "pointers", "docs")):
        assert _fill(got["text"][i], got["line_start"][i], ids) == _expect(w[key], offs, ids), key
    # no gap asked for: the values lines alone
    got = ix.query_text(capi.SPX_MODE_PML, seqs, offs, None, capi.SPX_TEXT_LENGTHS)
    assert got["text"][0] == b"".join(b"".join(b"%d " % int(v) for v in lens[offs[q]: offs[q + 1]]) + b"\n"
                                      for q in range(offs.size - 1))


def test_text_of_long_values_and_wide_reads(oracle_mod):
    """values with 1..5 digits (a read that matches for 20 000 characters), a read longer than 65 535 characters
    (32-bit values on the device), and the wave boundary cases: reads of 63, 64, 65 and 128 characters."""
    rng = np.random.default_rng(3)
    base = rng.choice(np.array(DNA, dtype=np.uint8), size=70000)
    raw = synth.index_from_text(__import__("torch").from_numpy(base.copy()), doc_lengths=[30000, 40000])
    reads = [base[100:20100], base[5:68005], base[7:70], base[7:71], base[7:72], base[300:428], base[0:1]]
    offs = np.concatenate([[0], np.cumsum([r.size for r in reads])]).astype(np.int64)
    seqs = np.concatenate(reads)
    ids = [b"r%d" % q for q in range(len(reads))]
    gap = np.array([len(i) + 2 for i in ids], dtype=np.uint32)
    orc = oracle_mod.OracleIndex.from_raw(raw)
    lens = orc.pml(seqs, offs)
    assert lens.max() >= 10000
    ix = capi.Index.from_raw(raw, 0)
    got = ix.query_text(capi.SPX_MODE_PML, seqs, offs, gap, capi.SPX_TEXT_LENGTHS)
    assert _fill(got["text"][0], got["line_start"][0], ids) == _expect(lens, offs, ids)


def test_text_with_digestion(oracle_mod):
    """-m: digestion, walk and formatting without the digested reads or the values leaving the device; a read that
    digests to nothing shows as a record of header + newline."""
    raw, text = cases.real_case(61, 12000, DNA)
    rng = np.random.default_rng(4)
    seqs, offs = cases.reads_mixed(rng, text, DNA, 80, 400)
    k, w = 4, 11
    dseqs, doffs = oracle_mod.digest_batch(oracle_mod.DIGEST_PROMOTED, k, w, seqs, offs)
    # an index over the promoted alphabet: digest the text itself
    dtext = oracle_mod.digest(oracle_mod.DIGEST_PROMOTED, k, w, text)
    import torch

    rawd = synth.index_from_text(torch.from_numpy(np.asarray(dtext).copy()))
    orc = oracle_mod.OracleIndex.from_raw(rawd)
    lens = orc.pml(dseqs, doffs)
    ids = [b"q%d" % q for q in range(offs.size - 1)]
    gap = np.array([len(i) + 2 for i in ids], dtype=np.uint32)
    ix = capi.Index.from_raw(rawd, 0)
    got = ix.query_text(capi.SPX_MODE_PML, seqs, offs, gap, capi.SPX_TEXT_LENGTHS, digest=(capi.SPX_DIGEST_PROMOTED, k, w))
    assert _fill(got["text"][0], got["line_start"][0], ids) == _expect(lens, np.asarray(doffs, dtype=np.int64), ids)


def test_text_scratch_reserved_ahead_and_two_contexts_at_once(oracle_mod):
    """spx_query_text_reserve (round 5): the scratch of a begin / fetch pair allocated ahead -- too small, exact and generous
    hints all give the same bytes (a hint, not a limit), for PML + docs + class and for MS.  Then two query contexts of one
    index (spx_index_clone onto the same device) run begin / fetch from two threads at once, many times: every result is
    the oracle's -- the sizes and counters a begin waits for are published by a kernel into page-locked host memory, and the
    class records arrive with the text, at fetch."""
    import threading
    raw, text = cases.real_case(77, 12000, DNA, ndocs=4)
    rng = np.random.default_rng(77)
    seqs, offs = cases.reads_mixed(rng, text, DNA, 400, 300, [2])
    ids = [b"r%d" % q for q in range(offs.size - 1)]
    gap = np.array([len(i) + 2 for i in ids], dtype=np.uint32)
    orc = oracle_mod.OracleIndex.from_raw(raw)
    lens, docs = orc.pml(seqs, offs, want_docs=True)
    want = (_expect(lens, offs, ids), _expect(docs, offs, ids))
    f, a, b, s_ = oracle_mod.classify(lens, offs, 20, 6)
    ix = capi.Index.from_raw(raw, 0)
    streams = capi.SPX_TEXT_LENGTHS | capi.SPX_TEXT_DOCS

    def check(handle):
        got = handle.query_text(capi.SPX_MODE_PML, seqs, offs, gap, streams, classify=(20, 6))
        assert _fill(got["text"][0], got["line_start"][0], ids) == want[0]
        assert _fill(got["text"][2], got["line_start"][2], ids) == want[1]
        assert np.array_equal(got["class"]["above"], a) and np.array_equal(got["class"]["below"], b)

    for chars, reads, tb in ((16, 1, (1, 1, 1)), (int(offs[-1]), offs.size - 1, None), (4 * int(offs[-1]), 4 * offs.size, (10 ** 6, 0, 10 ** 6))):
        ix.reserve_text(capi.SPX_MODE_PML, chars, reads, streams, classify=True, text_bytes=tb)
        check(ix)
    with pytest.raises(Exception):
        ix.reserve_text(7, 100, 10)
    # MS
    ix.set_text(__import__("torch").from_numpy(text.copy()))
    ix.reserve_text(capi.SPX_MODE_MS, 2 * int(offs[-1]), 2 * offs.size, capi.SPX_TEXT_LENGTHS | capi.SPX_TEXT_POINTERS)
    w = orc.ms(seqs, offs, text=text)
    got = ix.query_text(capi.SPX_MODE_MS, seqs, offs, gap, capi.SPX_TEXT_LENGTHS | capi.SPX_TEXT_POINTERS)
    assert _fill(got["text"][0], got["line_start"][0], ids) == _expect(w["lengths"], offs, ids)
    assert _fill(got["text"][1], got["line_start"][1], ids) == _expect(w["pointers"], offs, ids)
    # two contexts, two threads; one of them sleeps on an event while it waits for the device ("blocking_sync"), one spins
    other = ix.clone(0)
    other.set_option("blocking_sync", 1)
    errors = []

    def worker(handle):
        try:
            for _ in range(25):
                check(handle)
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(h,)) for h in (ix, other)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors[:2]
