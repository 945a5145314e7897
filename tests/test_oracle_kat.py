"""Known-answer tests that PIN THE ORACLE (the reference ships no golden vectors,
SURVEY.md 8(c)): primitives against an expanded BWT string, the synthetic index
builder against a naive suffix sort, MS lengths against brute-force matching
statistics (a mathematical invariant independent of anybody's code)."""
import numpy as np
import pytest
import torch

from spumoni_amd import synth
from tests import brute


def _text(rng, n, letters):
    return np.asarray(letters, dtype=np.uint8)[rng.integers(0, len(letters), size=n)]


def _repetitive_text(rng, n, letters):
    base = _text(rng, max(4, n // 6), letters)
    parts = []
    while sum(p.size for p in parts) < n:
        p = base.copy()
        k = rng.integers(0, 3)
        for _ in range(k):
            p[rng.integers(0, p.size)] = letters[rng.integers(0, len(letters))]
        parts.append(p[rng.integers(0, p.size // 2):])
    return np.concatenate(parts)[:n]


CASES = [
    (1, 40, list(b"ACGT")),
    (2, 64, list(b"AC")),
    (3, 120, list(b"ACGTN")),
    (4, 200, [3, 4, 5, 6, 7, 90, 91]),
    (5, 33, list(b"A")),
    (6, 257, list(b"ACGT")),
]


@pytest.mark.parametrize("seed,n,letters", CASES)
def test_index_builder_matches_naive(seed, n, letters):
    rng = np.random.default_rng(seed)
    text = _repetitive_text(rng, n, letters)
    raw = synth.index_from_text(torch.from_numpy(text), doc_lengths=[n // 2, n - n // 2])
    t = text.tolist() + [0]
    sa, bwt = brute.naive_bwt(t)
    heads, lens = brute.runs_of(bwt)
    assert raw.heads.tolist() == heads
    assert raw.lens.tolist() == lens
    assert raw.n == n + 1
    # thresholds: a position inside (end of prev same-letter run, start] with minimal LCP
    lcp = brute.naive_lcp(t, sa)
    starts = np.cumsum([0] + lens[:-1]).tolist()
    last_end = {}
    for k, (c, s, ln) in enumerate(zip(heads, starts, lens)):
        if c in last_end:
            lo, hi = last_end[c] + 1, s
            th = int(raw.thr[k])
            assert lo <= th <= hi
            assert lcp[th] == min(lcp[lo : hi + 1])
            assert th == lo + int(np.argmin(lcp[lo : hi + 1]))  # first arg-min
        else:
            assert int(raw.thr[k]) == 0
        last_end[c] = s + ln - 1
    # samples: text position of the BWT character at run start / end
    for k, (s, ln) in enumerate(zip(starts, lens)):
        assert int(raw.ssa[k]) == (sa[s] - 1) % (n + 1)
        assert int(raw.esa[k]) == (sa[s + ln - 1] - 1) % (n + 1)
    # docs: number of document ends <= sample (last doc absorbs the terminator)
    ends = [n // 2, n + 1]
    for k in range(len(heads)):
        assert int(raw.doc_start[k]) == sum(1 for e in ends if e <= int(raw.ssa[k]))


@pytest.mark.parametrize("seed,n,letters", CASES)
def test_primitives_vs_expanded_bwt(oracle_mod, seed, n, letters):
    rng = np.random.default_rng(seed)
    text = _repetitive_text(rng, n, letters)
    raw = synth.index_from_text(torch.from_numpy(text))
    ix = oracle_mod.OracleIndex.from_raw(raw)
    t = text.tolist() + [0]
    _, bwt = brute.naive_bwt(t)
    bwt = [max(c, 1) for c in bwt]  # 0 -> TERMINATOR (ms_rle_string.hpp:250)
    N = len(bwt)
    assert ix.n == N
    alphabet = sorted(set(bwt)) + [2, 200]  # also absent letters
    for p in range(N):
        assert ix.at(p) == bwt[p]
    for c in alphabet:
        nc = bwt.count(c)
        for p in range(N + 1):
            assert ix.rank(p, c) == brute.rank_brute(bwt, p, c)
        for i in range(nc):
            assert ix.select(i, c) == brute.select_brute(bwt, i, c)
        # LF = F[c] + rank
        Fc = sum(1 for x in bwt if x < c)
        for p in range(0, N + 1, 3):
            assert ix.LF(p, c) == Fc + brute.rank_brute(bwt, p, c)
    # LF on the real BWT is the inverse suffix array step: LF(i) for bwt[i]=c
    sa = brute.naive_sa(t)
    isa = {s: i for i, s in enumerate(sa)}
    for i in range(N):
        assert ix.LF(i, bwt[i]) == isa[(sa[i] - 1) % N]


def _reads_for(rng, text, letters, nreads, maxlen):
    reads = []
    for _ in range(nreads):
        m = int(rng.integers(1, maxlen))
        kind = rng.integers(0, 3)
        if kind == 0 and text.size > m:
            s = int(rng.integers(0, text.size - m))
            rd = text[s : s + m].copy()
            for _ in range(int(rng.integers(0, 3))):
                rd[rng.integers(0, m)] = letters[rng.integers(0, len(letters))]
        elif kind == 1:
            rd = _text(rng, m, letters)
        else:
            rd = _text(rng, m, letters + [ord("Z")])  # a letter absent from the index
        reads.append(rd)
    offs = np.concatenate([[0], np.cumsum([r.size for r in reads])]).astype(np.int64)
    return np.concatenate(reads), offs


@pytest.mark.parametrize("seed,n,letters", CASES)
def test_ms_lengths_equal_bruteforce_matching_statistics(oracle_mod, seed, n, letters):
    rng = np.random.default_rng(100 + seed)
    text = _repetitive_text(rng, n, letters)
    raw = synth.index_from_text(torch.from_numpy(text), doc_lengths=[n // 3, n - n // 3])
    ix = oracle_mod.OracleIndex.from_raw(raw)
    seqs, offs = _reads_for(rng, text, letters, 40, 30)
    res = ix.ms(seqs, offs, want_docs=True, text=text)
    pml, pdocs = ix.pml(seqs, offs, want_docs=True)
    for q in range(offs.size - 1):
        rd = seqs[offs[q] : offs[q + 1]]
        want = brute.true_ms(text, rd)
        got = res["lengths"][offs[q] : offs[q + 1]].tolist()
        present = set(text.tolist())
        if all(int(c) in present for c in rd):
            assert got == want, (q, rd.tobytes())
            # pointers really point at an occurrence of the match
            ptrs = res["pointers"][offs[q] : offs[q + 1]]
            for i, (p, l) in enumerate(zip(ptrs.tolist(), want)):
                if l > 0:
                    assert text[p : p + l].tobytes() == rd[i : i + l].tobytes()
        else:
            # absent letters produce the fake pointer 0, after which the reference's continuation
            # shortcut may under-report (see tests/test_oracle_hypothesis.py); never over-reports
            assert all(g <= w for g, w in zip(got, want)), (q, rd.tobytes())
        # PML is a lower bound on MS (Ahmed et al.; PML never over-reports)
        assert (pml[offs[q] : offs[q + 1]] <= np.asarray(want)).all()
    # doc ids of MS pointers: for positions reached by a jump, the doc of the sample
    assert res["docs"].shape == res["pointers"].shape


def test_signed_char_quirk_bytes_ge_128(oracle_mod):
    """Appendix C1: bytes >= 128 never take the match branch -> PML stays 0 on them."""
    rng = np.random.default_rng(7)
    letters = [3, 4, 130, 131, 200]
    text = _repetitive_text(rng, 150, letters)
    raw = synth.index_from_text(torch.from_numpy(text))
    ix = oracle_mod.OracleIndex.from_raw(raw)
    rd = text[20:60].copy()
    offs = np.array([0, rd.size])
    pml = ix.pml(rd, offs)
    hi = rd >= 128
    assert (pml[hi] == 0).all()
    # low bytes that follow (to the right of) a low byte can still extend
    assert pml.max() >= 1


def test_classifier_bins(oracle_mod):
    # one bin when shorter than the bin width; last bin absorbs a short tail (Appendix C11)
    L = np.array([0, 5, 1] + [0] * 10, dtype=np.uint32)
    f, a, b, s = oracle_mod.classify(L, np.array([0, L.size]), 150, 5)
    assert (int(f[0]), int(a[0]), int(b[0]), int(s[0])) == (1, 1, 0, 5)
    L = np.zeros(449, dtype=np.uint32)
    L[10] = 9
    L[160] = 2
    L[448] = 7
    f, a, b, s = oracle_mod.classify(L, np.array([0, L.size]), 150, 7)
    # bins [0,150) [150,449): 449-300 < 150 so the second bin runs to the end
    assert (int(a[0]), int(b[0]), int(s[0]), int(f[0])) == (2, 0, 16, 1)
    L = np.zeros(450, dtype=np.uint32)
    L[449] = 7
    f, a, b, s = oracle_mod.classify(L, np.array([0, L.size]), 150, 7)
    assert (int(a[0]), int(b[0]), int(f[0])) == (1, 2, 0)
    assert oracle_mod.max_value_thr(2.0, True, False, False) == 7
    assert oracle_mod.max_value_thr(9.7, True, True, False) == 9
    assert oracle_mod.max_value_thr(9.7, True, False, True) == 10
    assert oracle_mod.max_value_thr(9.7, False, False, False) == 9
