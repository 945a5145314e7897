"""Property-based pinning of the oracle (hypothesis): for arbitrary small texts and reads,
MS lengths from the oracle (pointers + extension) equal brute-force matching statistics, every
pointer points at an occurrence, PML never exceeds MS, and primitives agree with the expanded BWT."""
import numpy as np
import torch
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from spumoni_amd import synth
from tests import brute

ALPHABETS = [b"AC", b"ACGT", b"ACGTN", bytes([3, 4, 5, 90, 100])]


@st.composite
def text_and_reads(draw):
    alpha = draw(st.sampled_from(ALPHABETS))
    n = draw(st.integers(min_value=1, max_value=60))
    text = bytes(draw(st.lists(st.sampled_from(list(alpha)), min_size=n, max_size=n)))
    pool = list(alpha) + [ord("Z")]
    reads = draw(st.lists(st.lists(st.sampled_from(pool), min_size=1, max_size=25).map(bytes), min_size=1, max_size=6))
    # make some reads genuine substrings so that long matches occur
    k = draw(st.integers(min_value=0, max_value=max(0, n - 1)))
    ln = draw(st.integers(min_value=1, max_value=n - k))
    reads.append(text[k : k + ln])
    return text, reads


@settings(max_examples=400, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(text_and_reads())
def test_ms_equals_bruteforce(oracle_mod, tr):
    text_b, reads = tr
    text = np.frombuffer(text_b, dtype=np.uint8)
    raw = synth.index_from_text(torch.from_numpy(text.copy()))
    orc = oracle_mod.OracleIndex.from_raw(raw)
    seqs = np.frombuffer(b"".join(reads), dtype=np.uint8)
    offs = np.concatenate([[0], np.cumsum([len(r) for r in reads])])
    ms = orc.ms(seqs, offs, text=text)
    pml = orc.pml(seqs, offs)
    for q, rd in enumerate(reads):
        want = brute.true_ms(text_b, rd)
        got = ms["lengths"][offs[q] : offs[q + 1]].tolist()
        if all(c in text_b for c in rd):
            assert got == want
            for i, (p, l) in enumerate(zip(ms["pointers"][offs[q] : offs[q + 1]].tolist(), want)):
                if l:
                    assert text_b[p : p + l] == rd[i : i + l]
        else:
            # A letter absent from the index sets the pointer to 0 (compute_ms_pml.cpp:581); if the
            # next position's pointer happens to be 1 the extension loop takes it for a continuation
            # (`pos != pointers[i-1] + 1`, :805) and under-reports.  That is the reference's
            # behaviour (found by this test: text AAAC, read ZAA -> 0 0 0, true MS 0 2 1); the
            # restatement keeps it, so here lengths are only bounded by the true MS.
            assert all(g <= w for g, w in zip(got, want))
        assert (pml[offs[q] : offs[q + 1]] <= np.asarray(want)).all()


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(st.lists(st.sampled_from(list(b"ACGT")), min_size=1, max_size=40).map(bytes))
def test_primitives(oracle_mod, text_b):
    text = np.frombuffer(text_b, dtype=np.uint8)
    raw = synth.index_from_text(torch.from_numpy(text.copy()))
    orc = oracle_mod.OracleIndex.from_raw(raw)
    _, bwt = brute.naive_bwt(list(text_b) + [0])
    bwt = [max(c, 1) for c in bwt]
    for c in set(bwt) | {ord("Z")}:
        for p in range(len(bwt) + 1):
            assert orc.rank(p, c) == brute.rank_brute(bwt, p, c)
        for i in range(bwt.count(c)):
            assert orc.select(i, c) == brute.select_brute(bwt, i, c)
