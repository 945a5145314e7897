"""The builder's null database (tooling): null reads chosen like src/refbuilder.cpp:83-127 / :234-270 chooses
them (glibc's rand() after srand(0)), the one-sided KS statistic of src/ks_test.cpp:58-134 and the threshold
mean + 3 sd of compute_ms_pml.cpp:1549-1663, the file of src/emp_null_database.cpp:83-101.  CPU only: the generator
against the C library's own, the statistic against scipy, the windows and the file by hand."""
import ctypes
import struct

import numpy as np
import pytest

from spumoni_amd import build_index as B


def _libc():
    try:
        return ctypes.CDLL("libc.so.6")
    except OSError:
        pytest.skip("no glibc to compare with")


@pytest.mark.parametrize("seed", [0, 1, 2, 42, 2**31 - 1, 2**31 + 5, 2**32 - 1])
def test_rand_is_glibc_rand(seed):
    libc = _libc()
    libc.srand(ctypes.c_uint(seed))
    want = [libc.rand() for _ in range(2000)]
    g = B.GlibcRand(seed)
    assert [g.rand() for _ in range(2000)] == want
    assert B.GlibcRand(0).rand() == B.GlibcRand(1).rand() == 1804289383  # srand(0) is srand(1)


def _seqs(rng, lens):
    return [rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n) for n in lens]


def test_null_reads_of_a_file_list_follow_the_draws():
    libc = _libc()
    rng = np.random.default_rng(1)
    files = [_seqs(rng, [5000, 120, 151]), _seqs(rng, [900])]
    reads = B.null_reads_from_list(files, B.GlibcRand(0))
    libc.srand(0)
    want = []
    for s in (s for f in files for s in f):
        if s.size > 150:
            for _ in range(100):
                at = libc.rand() % (s.size - 150)
                want.append(s[at: at + 150])
        else:
            want.append(s)  # at most 150 characters: the whole sequence, once
    assert len(reads) == len(want) == 100 + 1 + 100 + 100
    assert all(np.array_equal(a, b) for a, b in zip(reads, want))
    assert reads[100].size == 120 and all(r.size == 150 for i, r in enumerate(reads) if i != 100)


def test_null_reads_stop_at_the_bound_and_thin_out_after_800():
    rng = np.random.default_rng(2)
    files = [_seqs(rng, [400] * 14)]
    reads = B.null_reads_from_list(files, B.GlibcRand(0))
    # 8 sequences x 100 = 800, then 25 per sequence: 8 more sequences would give 200 -- the bound of 1000 holds,
    # and with 14 sequences there are 800 + 6 x 25 = 950
    assert len(reads) == 950
    assert len(B.null_reads_from_list([_seqs(rng, [400] * 40)], B.GlibcRand(0))) == 1000
    # short sequences are added whatever the count is (the reference does not test the bound there)
    assert len(B.null_reads_from_list([_seqs(rng, [400] * 40 + [30, 30])], B.GlibcRand(0))) == 1002


def test_null_reads_of_a_single_fasta_drop_pieces_with_N_but_spend_the_draw():
    libc = _libc()
    rng = np.random.default_rng(3)
    s = _seqs(rng, [3000])[0].copy()
    s[1000:1400] = ord("N")
    s2 = _seqs(rng, [100])[0]
    reads = B.null_reads_from_fasta([s, s2], B.GlibcRand(0))
    libc.srand(0)
    want = []
    for _ in range(100):
        at = libc.rand() % (3000 - 150)
        piece = s[at: at + 150]
        if not (piece == ord("N")).any():
            want.append(piece)
    want.append(s2)
    assert 40 < len(want) < 101
    assert len(reads) == len(want) and all(np.array_equal(a, b) for a, b in zip(reads, want))
    # lower-case n does not count as N (the file is not upper-cased on this path: src/refbuilder.cpp:251-255)
    low = s.copy()
    low[low == ord("N")] = ord("n")
    assert len(B.null_reads_from_fasta([low], B.GlibcRand(0))) == 100
    # reading stops with the sequence that reaches 1000
    many = _seqs(rng, [500] * 30)
    assert len(B.null_reads_from_fasta(many, B.GlibcRand(0))) == 1000


def test_ks_statistic_is_the_one_sided_two_sample_statistic():
    stats = pytest.importorskip("scipy.stats")
    rng = np.random.default_rng(4)
    for trial in range(200):
        a = rng.integers(0, int(rng.integers(1, 40)), size=int(rng.integers(1, 300)))
        b = rng.integers(0, int(rng.integers(1, 40)), size=int(rng.integers(1, 300)))
        if trial % 3 == 0:
            a = a + int(rng.integers(0, 10))  # positives shifted to the right
        got = B.ks_statistic(a, b)
        # sup_x (F_null(x) - F_pos(x)); scipy calls it the statistic of alternative="less" for (pos, null)
        want = stats.ks_2samp(a, b, alternative="less", method="asymp").statistic
        assert abs(got - want) < 1e-12, (trial, got, want)
    assert B.ks_statistic([3, 3, 3], [3, 3]) == 0.0
    assert B.ks_statistic([9], [0]) == 1.0
    assert B.ks_statistic([0], [9]) == 0.0  # positives to the LEFT of the null do not count (src/ks_test.cpp:96-100)


def test_ks_statistic_like_the_loop_of_the_reference():
    """The loop itself (sort, CDFs over 0..max, stop after the first value at which a CDF is 1), written out."""
    rng = np.random.default_rng(5)
    for _ in range(100):
        pos = sorted(rng.integers(0, 12, size=int(rng.integers(1, 60))).tolist())
        nul = sorted(rng.integers(0, 12, size=int(rng.integers(1, 60))).tolist())
        top = max(pos[-1], nul[-1])

        def cdf(v):
            out, at = [], 0
            for x in range(top + 1):
                while at < len(v) and v[at] == x:
                    at += 1
                out.append(at / (len(v) + 0.0))
            return out

        ks = 0.0
        for p, q in zip(cdf(pos), cdf(nul)):
            ks = max(q - p, ks)
            if p >= 1.0 or q >= 1.0:
                break
        assert B.ks_statistic(pos, nul) == ks


class _Draws:
    def __init__(self, vals):
        self.vals, self.n = list(vals), 0

    def rand(self):
        self.n += 1
        return self.vals[(self.n - 1) % len(self.vals)]


def test_windows_of_a_read():
    null = np.arange(1000) % 7
    # shorter than a bin: one window of the whole read; exactly one draw per window
    d = _Draws([5])
    assert len(B.run_kstest(np.ones(40, dtype=np.int64), null, 150, d)) == 1 and d.n == 1
    # 150: start + 150 <= 150 - 150 fails -> the window runs to the end; 299 the same; 300 -> 150 + 150; 449 -> 150 + 299
    for m, want in ((150, 1), (299, 1), (300, 2), (449, 2), (450, 3)):
        d = _Draws([5])
        assert len(B.run_kstest(np.ones(m, dtype=np.int64), null, 150, d)) == want and d.n == want, m
    # the null window starts at rand() % (num_values - 2 * bin) and is as long as the read's window
    pos = np.full(200, 3)
    for draw in (0, 123, 699, 700, 1403):
        at = draw % (1000 - 300)
        assert B.run_kstest(pos, null, 150, _Draws([draw])) == [B.ks_statistic(pos, null[at: at + 200])]
    # fewer null statistics than two bins: the window starts at 0
    small = np.arange(100) % 5
    assert B.run_kstest(np.full(60, 2), small, 150, _Draws([77])) == [B.ks_statistic(np.full(60, 2), small[:60])]
    with pytest.raises(ValueError):
        B.run_kstest(pos, np.zeros(300, dtype=np.int64), 150, _Draws([1]))


def test_threshold_is_mean_plus_three_sd_of_all_windows():
    rng = np.random.default_rng(6)
    null = rng.integers(0, 9, size=3000)
    reads = [rng.integers(0, 9, size=int(n)) for n in (150, 150, 40, 0, 320)]
    thr = B.ks_threshold(reads, null, 150, B.GlibcRand(0))
    g = B.GlibcRand(0)
    ks = [k for rd in reads if rd.size for k in B.run_kstest(rd, null, 150, g)]
    assert len(ks) == 5  # 1 + 1 + 1 + (empty: none) + 2
    assert thr == pytest.approx(np.mean(ks) + 3 * np.std(ks), rel=1e-12)
    assert 0 < thr < 1


def test_null_db_file_fields_and_width_rule(tmp_path):
    p = str(tmp_path / "x.pmlnulldb")
    stats = [1, 2, 2, 2, 2, 2, 5, 5, 5, 5, 5, 7, 8]  # largest 8: ceil(log2(8)) = 3 bits, 8 is stored as 0 (as upstream)
    B.write_null_db(p, stats, 0.1375)
    blob = open(p, "rb").read()
    num, ks, mean, pct = struct.unpack("<Qddd", blob[:32])
    assert (num, ks, pct) == (13, 0.1375, 5.0) and mean == pytest.approx(np.mean(stats))
    bits, width = struct.unpack("<QB", blob[32:41])
    assert (bits, width) == (13 * 3, 3)
    word = struct.unpack("<Q", blob[41:49])[0]
    assert [(word >> (3 * i)) & 7 for i in range(13)] == [v & 7 for v in stats]
    assert B._width([0]) == 1 and B._width([1]) == 1 and B._width([2]) == 1 and B._width([3]) == 2 and B._width([9]) == 4


def test_fasta_reader_takes_gzip_like_gzopen(tmp_path):
    import gzip

    body = b">x some words\nacgt\nAC\n\n>y\nGG\n>empty\n"
    (tmp_path / "a.fa").write_bytes(body)
    with gzip.open(tmp_path / "b.fa.gz", "wb") as f:
        f.write(body)
    for name in ("a.fa", "b.fa.gz"):
        assert [s.tobytes() for s in B.read_fasta(str(tmp_path / name))] == [b"ACGTAC", b"GG"]
        assert [s.tobytes() for s in B.read_fasta(str(tmp_path / name), upper=False)] == [b"acgtAC", b"GG"]
