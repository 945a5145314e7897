"""Tier T1 of the oracle (Elias-Fano + Huffman wavelet tree + B-run block walk, the structures
of ri::rle_string / thr_bv) must agree with the flat tier T2 bit for bit, and with the expanded
BWT on tiny texts.  Two structurally independent restatements of the same reference semantics."""
import numpy as np
import pytest
import torch

from spumoni_amd import synth
from tests import brute, cases


@pytest.mark.parametrize("seed,n,letters", [(1, 60, list(b"ACGT")), (2, 200, list(b"AC")), (3, 150, [3, 4, 90, 128, 200, 255]),
                                            (4, 40, [ord("A")]), (5, 300, list(range(3, 40)))])
def test_t1_primitives_vs_expanded_bwt(oracle_mod, seed, n, letters):
    rng = np.random.default_rng(seed)
    text = cases.repetitive_text(rng, n, letters)
    raw = synth.index_from_text(torch.from_numpy(text))
    t1 = oracle_mod.OracleT1Index.from_raw(raw)
    t2 = oracle_mod.OracleIndex.from_raw(raw)
    _, bwt = brute.naive_bwt(text.tolist() + [0])
    bwt = [max(c, 1) for c in bwt]
    N = len(bwt)
    for p in range(N):
        assert t1.at(p) == bwt[p]
        assert t1.run_of_position(p) == t2.run_of_position(p)
    for c in sorted(set(bwt)) + [2, 250]:
        for p in range(N + 1):
            assert t1.rank(p, c) == brute.rank_brute(bwt, p, c)
        for i in range(bwt.count(c)):
            assert t1.select(i, c) == brute.select_brute(bwt, i, c)
    for k in range(raw.r):
        assert t1.threshold(k) == t2.threshold(k)


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_t1_equals_t2_on_real_bwt(oracle_mod, seed):
    letters = [list(b"ACGT"), list(b"ACGTN"), [3, 4, 5, 90, 127, 128, 129, 200, 255]][seed - 11]
    raw, text = cases.real_case(seed, 4000, letters)
    rng = np.random.default_rng(seed)
    seqs, offs = cases.reads_mixed(rng, text, letters, 200, 100, [ord("Z"), 2])
    t1 = oracle_mod.OracleT1Index.from_raw(raw)
    t2 = oracle_mod.OracleIndex.from_raw(raw)
    a, ad = t1.pml(seqs, offs, want_docs=True)
    b, bd = t2.pml(seqs, offs, want_docs=True)
    assert np.array_equal(a, b) and np.array_equal(ad, bd)
    m1, m2 = t1.ms(seqs, offs, want_docs=True), t2.ms(seqs, offs, want_docs=True)
    assert np.array_equal(m1["pointers"], m2["pointers"]) and np.array_equal(m1["docs"], m2["docs"])


def test_t1_equals_t2_on_statistical_index(oracle_mod):
    raw = synth.statistical_rlbwt(20000, 253, 4.0, seed=3, zipf=1.0, with_samples=True, n_docs=7)
    seqs, offs = synth.simulate_reads(raw, 1500, 44, seed=5)
    t1 = oracle_mod.OracleT1Index.from_raw(raw)
    t2 = oracle_mod.OracleIndex.from_raw(raw)
    assert np.array_equal(t1.pml(seqs.numpy(), offs.numpy()), t2.pml(seqs.numpy(), offs.numpy()))
    m1, m2 = t1.ms(seqs.numpy(), offs.numpy(), want_docs=True), t2.ms(seqs.numpy(), offs.numpy(), want_docs=True)
    assert np.array_equal(m1["pointers"], m2["pointers"]) and np.array_equal(m1["docs"], m2["docs"])
