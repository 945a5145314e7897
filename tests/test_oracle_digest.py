"""Digestion oracle (oracle/orc_digest.c) against the slice-and-min specification in brute.py,
hand-worked cases and the pieces that can be checked independently (MT19937 against numpy's)."""
import numpy as np
import pytest

from tests import brute


def _rand_read(rng, n, p_n=0.03):
    s = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n)
    s[rng.random(n) < p_n] = ord("N")
    return s.tobytes()


def test_default_charhash_is_mt19937_1337(oracle_mod):
    bg = np.random.MT19937()
    bg._legacy_seeding(1337)  # init_genrand(1337)
    raw = bg.random_raw(256)
    want = [int(raw[ord(c)] & 0xFF) for c in "ACGT"]
    assert list(oracle_mod.digest_default_charhash()) == want


def test_hand_worked_dna(oracle_mod):
    # k=2, w=3: two 2-mers per window, ordered by code ^ (XOR_MASK & 15) = code ^ 13
    # AC=1->12  CG=6->11  GT=11->6  TA=12->1  AA=0->13
    # read ACGTAA: 2-mers AC CG GT TA AA; windows (AC,CG)->CG (CG,GT)->GT (GT,TA)->TA (TA,AA)->TA
    got = bytes(oracle_mod.digest(oracle_mod.DIGEST_DNA, 2, 3, b"ACGTAA"))
    assert got == b"CGGTTA"
    # w == k: every k-mer reports itself, consecutive duplicates dropped
    assert bytes(oracle_mod.digest(oracle_mod.DIGEST_DNA, 1, 1, b"AACCCGT")) == b"ACGT"
    # a character outside ACGT restarts the k-mer but the window keeps its contents
    assert bytes(oracle_mod.digest(oracle_mod.DIGEST_DNA, 2, 3, b"ACNGT")) == b"GT"
    # too short for one full window: nothing
    assert len(oracle_mod.digest(oracle_mod.DIGEST_DNA, 4, 11, b"ACGTACGTAC")) == 0
    assert len(oracle_mod.digest(oracle_mod.DIGEST_DNA, 4, 11, b"ACGTACGTACG")) == 4
    assert len(oracle_mod.digest(oracle_mod.DIGEST_DNA, 4, 11, b"")) == 0


def test_hand_worked_promoted(oracle_mod):
    ch = [1, 2, 4, 8]  # T[A], T[C], T[G], T[T]
    # k=1, w=1: every base reports its own hash; 1 and 2 are promoted past PFP's reserved bytes (+3)
    got = list(oracle_mod.digest(oracle_mod.DIGEST_PROMOTED, 1, 1, b"ACGTT", ch))
    assert got == [4, 5, 4, 8]
    # k=2: h = rotl8(T[c0], 1) ^ T[c1]; AC -> 2^2=0 -> promoted to 3; CA -> 4^1=5
    got = list(oracle_mod.digest(oracle_mod.DIGEST_PROMOTED, 2, 2, b"ACA", ch))
    assert got == [3, 5]
    # window of two: min(0, 5) = 0, then min(5, 0) = 0 again -> one value
    got = list(oracle_mod.digest(oracle_mod.DIGEST_PROMOTED, 2, 3, b"ACAC", ch))
    assert got == [3]


@pytest.mark.parametrize("kind", [1, 2])
def test_against_specification(oracle_mod, kind):
    rng = np.random.default_rng(5 + kind)
    default = list(oracle_mod.digest_default_charhash())
    for trial in range(300):
        k = int(rng.integers(1, 5))
        w = k + int(rng.integers(0, 12))
        n = int(rng.integers(0, 300))
        read = _rand_read(rng, n, p_n=float(rng.choice([0.0, 0.02, 0.2])))
        ch = default if trial % 2 == 0 else [int(x) for x in rng.integers(0, 256, size=4)]
        want = brute.digest_spec(kind, k, w, read, ch)
        got = bytes(oracle_mod.digest(kind, k, w, read, None if trial % 2 == 0 else ch))
        assert got == want, (k, w, read)


def test_batch_form_and_density(oracle_mod):
    rng = np.random.default_rng(9)
    reads = [_rand_read(rng, int(n), 0.0) for n in rng.integers(0, 400, size=50)]
    seqs = np.frombuffer(b"".join(reads), dtype=np.uint8)
    offs = np.cumsum([0] + [len(r) for r in reads]).astype(np.uint64)
    out, out_offs = oracle_mod.digest_batch(1, 4, 11, seqs, offs)
    for q, r in enumerate(reads):
        assert bytes(out[int(out_offs[q]) : int(out_offs[q + 1])]) == bytes(oracle_mod.digest(1, 4, 11, r))
    # the promoted alphabet never uses PFP's reserved bytes 0, 1, 2
    assert out.min() >= 3
    # density of window minimizers on random DNA is about 2 / (wsz + 1) = 0.22 for k=4, w=11
    long_read = _rand_read(rng, 200000, 0.0)
    d = len(oracle_mod.digest(2, 4, 11, long_read)) / 4 / len(long_read)
    assert 0.15 < d < 0.30
