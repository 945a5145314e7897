"""-m gpu: the flat-layout cache (.spx), index replication, and the text-against-index check.

pml_t / ms_t deserialise their index on every run (compute_ms_pml.cpp:700-721, 755-786); here the
flat arrays are written once and brought back by plain copies, and an index is replicated to the
other devices by a device-to-device copy instead of being flattened again."""
import filecmp

import numpy as np
import pytest
import torch

from spumoni_amd import capi, synth
from tests import cases
from tests.test_gpu_parity import _compare_all

pytestmark = pytest.mark.gpu

DNA = list(b"ACGT")


@pytest.mark.parametrize("kind", ["real_ms_doc", "statistical_pml", "wide_rows"])
def test_cache_round_trip_is_byte_identical_and_queries_agree(oracle_mod, tmp_path, kind, monkeypatch):
    rng = np.random.default_rng(5)
    if kind == "wide_rows":
        monkeypatch.setenv("SPX_ROWS_WIDE", "1")
    if kind == "statistical_pml":
        raw = synth.statistical_rlbwt(40_000, 60, 4.0, seed=3, device="cuda", zipf=1.0)
        text = None
        seqs, offs = synth.simulate_reads(raw, 2000, 50, seed=4)
        seqs, offs = seqs.cpu().numpy(), offs.cpu().numpy()
    else:
        raw, text = cases.real_case(31, 8000, DNA, ndocs=4)
        seqs, offs = cases.reads_mixed(rng, text, DNA, 300, 150, [ord("N")])
    fresh = capi.Index.from_raw(raw, 0)
    a, b = str(tmp_path / "a.spx"), str(tmp_path / "b.spx")
    fresh.save(a)
    loaded = capi.Index.load_flat(a, 0)
    assert (loaded.n, loaded.r) == (fresh.n, fresh.r)
    assert loaded.describe() == fresh.describe()
    loaded.save(b)
    assert filecmp.cmp(a, b, shallow=False), "cache of a loaded index differs from the cache of the fresh one"
    # a second flatten of the same input writes the same bytes (the layout is a function of the input)
    again = capi.Index.from_raw(raw, 0)
    again.save(b)
    assert filecmp.cmp(a, b, shallow=False), "flattening is not deterministic"
    _compare_all(oracle_mod, raw, text, seqs, offs, ix=loaded)


def test_clone_answers_like_the_original(oracle_mod):
    raw, text = cases.real_case(32, 6000, [3, 4, 90, 128, 200, 255], ndocs=3)
    rng = np.random.default_rng(6)
    seqs, offs = cases.reads_mixed(rng, text, [3, 4, 90, 128, 200, 255], 200, 120, [2])
    src = capi.Index.from_raw(raw, 0)
    dup = src.clone(0)
    assert dup.describe() == src.describe()
    src.close()  # the clone owns its arrays
    _compare_all(oracle_mod, raw, text, seqs, offs, ix=dup)


def test_cache_of_another_layout_is_refused(tmp_path):
    raw = synth.statistical_rlbwt(500, 5, 3.0, seed=1)
    ix = capi.Index.from_raw(raw, 0)
    p = str(tmp_path / "x.spx")
    ix.save(p)
    blob = bytearray(open(p, "rb").read())
    blob[8:12] = b"old!"  # the layout tag follows the 8-byte magic
    open(p, "wb").write(bytes(blob))
    with pytest.raises(capi.SpxError, match="rebuild the cache"):
        capi.Index.load_flat(p, 0)
    open(p, "wb").write(b"not a cache")
    with pytest.raises(capi.SpxError):
        capi.Index.load_flat(p, 0)
    with pytest.raises(capi.SpxError):
        capi.Index.load_flat(str(tmp_path / "missing.spx"), 0)


def test_text_that_is_not_the_indexed_text_is_refused():
    """a wrong SPUMONI_TEXT silently gave wrong .lengths.  The text is now checked against
    the index: its length, and text[samples_start[k]] == head of run k for every run."""
    raw, text = cases.real_case(33, 5000, DNA, ndocs=2)
    t = torch.from_numpy(text.copy())
    raw.text = None
    ix = capi.Index.from_raw(raw, 0)
    ix.set_text(t)  # the right text passes
    with pytest.raises(capi.SpxError, match="characters"):
        ix.set_text(t[:-1])  # wrong length
    # (the check looks at one text position per run -- the r positions the SA samples name -- so it
    # catches another text, not every local edit of the right one)
    rev = torch.flip(t, [0])  # same length, same letters, another text (e.g. missing reverse complements)
    with pytest.raises(capi.SpxError, match="disagrees"):
        ix.set_text(rev)
    with pytest.raises(capi.SpxError):  # after a refusal the index has no text
        ix.query_host(capi.SPX_MODE_MS, text[:50], np.array([0, 50]))
    ix.set_text(rev, unchecked=True)  # explicit opt-out (synthetic indexes)
    ix.set_text(t)
    got = ix.query_host(capi.SPX_MODE_MS, text[100:180], np.array([0, 80]))
    assert got["lengths"][0] >= 80 - 0  # a substring of the text matches to its end


@pytest.mark.parametrize("seed,letters,wide", [(41, DNA, 0), (42, [3, 4, 5, 90, 127, 128, 129, 200, 255], 0), (43, DNA + [ord("N")], 1)])
def test_text_rebuilt_from_the_index_is_the_text(oracle_mod, seed, letters, wide, monkeypatch):
    """ms_t needs the indexed text (through an SLP upstream); spx_index_rebuild_text recovers it from the MS index
    itself -- LF chains from every run's SA sample.  It must be the text byte for byte, and MS lengths computed with
    it must be the oracle's."""
    if wide:
        monkeypatch.setenv("SPX_ROWS_WIDE", "1")
    raw, text = cases.real_case(seed, 7000, letters, ndocs=3)
    raw.text = None
    ix = capi.Index.from_raw(raw, 0)
    ix.rebuild_text()
    assert np.array_equal(ix.text(), text)
    rng = np.random.default_rng(seed)
    seqs, offs = cases.reads_mixed(rng, text, letters, 200, 150, [2])
    _compare_all(oracle_mod, raw, text, seqs, offs, ix=ix)
    pml_only = capi.Index.from_raw(synth.RawIndex(heads=raw.heads, lens=raw.lens, thr=raw.thr, n=raw.n), 0)
    with pytest.raises(capi.SpxError, match="SA samples"):
        pml_only.rebuild_text()


def test_source_tag_travels_with_cache_and_clone(tmp_path):
    """nothing tied a .spx cache to the index files it was written for.  The caller's fingerprint of
    those files is saved in the header and handed back on load (the CLI refuses a cache whose tag differs)."""
    raw = synth.statistical_rlbwt(3000, 20, 4.0, seed=2, device="cuda")
    ix = capi.Index.from_raw(raw, 0)
    assert ix.source_tag() == ""
    ix.set_source_tag("pml:3 files:0123456789abcdef")
    p = str(tmp_path / "t.spx")
    ix.save(p)
    assert capi.Index.load_flat(p, 0).source_tag() == "pml:3 files:0123456789abcdef"
    assert ix.clone(0).source_tag() == "pml:3 files:0123456789abcdef"
    ix.set_source_tag("x" * 500)  # cut to what the header holds
    assert ix.source_tag() == "x" * 127


def test_damaged_cache_header_is_refused(tmp_path):
    """spx_index_load_flat trusted the header's array sizes / offsets; a damaged file could size device
    arrays the kernels then run past.  The fields are checked against each other and against the file."""
    import struct

    raw = synth.statistical_rlbwt(2000, 12, 4.0, seed=5, device="cuda", with_samples=True, n_docs=3)
    ix = capi.Index.from_raw(raw, 0)
    p = str(tmp_path / "d.spx")
    ix.save(p)
    good = open(p, "rb").read()
    assert capi.Index.load_flat(p, 0).r == ix.r
    # header: magic 8, layout 56, header_bytes 8, n 8, r 8, has_samples 4, has_docs 4, n_text 8, arr_bytes[10] ...
    off_r, off_arr = 8 + 56 + 8 + 8, 8 + 56 + 8 + 8 + 8 + 4 + 4 + 8
    for name, (at, val) in {"r": (off_r, ix.r + 1000), "rows bytes": (off_arr, 64), "dirrows bytes": (off_arr + 8, 1 << 40),
                            "rows offset": (off_arr + 80, len(good) + 4096)}.items():
        blob = bytearray(good)
        blob[at:at + 8] = struct.pack("<Q", val)
        open(p, "wb").write(bytes(blob))
        with pytest.raises(capi.SpxError, match="consistent"):
            capi.Index.load_flat(p, 0)
    open(p, "wb").write(good[: len(good) // 2])  # truncated
    with pytest.raises(capi.SpxError):
        capi.Index.load_flat(p, 0)


def test_long_runs_are_laid_out_as_pieces(oracle_mod, tmp_path, monkeypatch):
    """A run of 2^16 positions or more used to switch the whole index to the general row encoding (17-33 % slower).
    It is now laid out as pieces of the same head (spx_flatten.hip): compact rows stay, the API reports the file's r,
    every mode answers like the oracle, the text is rebuilt through the pieces, the cache round-trips -- and
    SPX_NO_PIECES=1 (the general encoding) gives the same answers."""
    rng = np.random.default_rng(12)
    chunks = [rng.choice(np.array(DNA, dtype=np.uint8), size=3000), np.full(150_000, ord("A"), dtype=np.uint8),
              rng.choice(np.array(DNA, dtype=np.uint8), size=2500), np.full(70_000, 200, dtype=np.uint8),  # a byte >= 128
              rng.choice(np.array(DNA, dtype=np.uint8), size=2000)]
    text = np.concatenate(chunks)
    raw = synth.index_from_text(torch.from_numpy(text.copy()), doc_lengths=[100_000, text.size - 100_000])
    assert int(raw.lens.max()) >= 65536
    reads = [text[2000:4500], text[150_000:156_000], np.full(3000, ord("A"), dtype=np.uint8), text[155_000:159_500],
             np.full(500, 200, dtype=np.uint8), rng.choice(np.array(DNA + [200], dtype=np.uint8), size=800)]
    offs = np.concatenate([[0], np.cumsum([r.size for r in reads])]).astype(np.int64)
    seqs = np.concatenate(reads)
    ix = capi.Index.from_raw(raw, 0)
    d = ix.describe()
    assert d["compact_rows"] == 1 and d["flat_runs"] > d["r"] == raw.r == ix.r
    _compare_all(oracle_mod, raw, text, seqs, offs, ix=ix)
    ix.rebuild_text()
    assert np.array_equal(ix.text(), text)
    p = str(tmp_path / "pieces.spx")
    ix.save(p)
    back = capi.Index.load_flat(p, 0)
    assert back.describe() == d and back.r == raw.r
    _compare_all(oracle_mod, raw, text, seqs, offs, ix=back)
    monkeypatch.setenv("SPX_NO_PIECES", "1")
    gen = capi.Index.from_raw(raw, 0)
    assert gen.describe()["compact_rows"] == 0 and gen.describe()["flat_runs"] == raw.r
    _compare_all(oracle_mod, raw, text, seqs, offs, ix=gen)
