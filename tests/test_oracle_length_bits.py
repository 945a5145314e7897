"""The identity the HIP path's PML output rests on, checked on the ORACLE's own outputs (CPU only).

The plain walk no longer writes a length per character: it writes one bit per character -- "the length was reset
here" -- and k_expand_lengths rebuilds the vector as the distance to the next set bit at or after each position (the
read's end when there is none).  That is exact iff, in every vector pml_pointers::_query can produce
(compute_ms_pml.cpp:238-286: `length = 0` on an absent letter or a jump, `length++` otherwise, walking the read from
its last character to its first), lengths[p] == q - p with q = min{ j >= p : lengths[j] == 0 } (or m).  Here the
oracle's vectors are encoded and decoded that way in numpy and compared with themselves."""
import numpy as np
import pytest

from tests import cases

DNA = list(b"ACGT")


def lengths_from_bits(bits):
    """bits[p] = 1 where the length was reset; the vector k_expand_lengths writes for one read"""
    m = bits.size
    nxt = np.where(bits, np.arange(m), m)
    nxt = np.minimum.accumulate(nxt[::-1])[::-1]  # first set position at or after p, else m
    return (nxt - np.arange(m)).astype(np.uint64)


@pytest.mark.parametrize("seed,letters,extra", [(5, DNA, [ord("N")]), (6, [3, 4, 90, 128, 200, 255], [2]), (7, list(b"AC"), [])])
def test_pml_lengths_are_a_function_of_their_zero_positions(oracle_mod, seed, letters, extra):
    raw, text = cases.real_case(seed, 6000, letters, ndocs=2)
    rng = np.random.default_rng(seed)
    seqs, offs = cases.reads_mixed(rng, text, letters, 400, 300, extra)
    orc = oracle_mod.OracleIndex.from_raw(raw)
    want = orc.pml(seqs, offs)
    assert want.max() > 5 and (want == 0).any()  # (bytes >= 128 never match: short lengths with that alphabet)
    for q in range(offs.size - 1):
        v = want[offs[q]:offs[q + 1]]
        assert np.array_equal(lengths_from_bits(v == 0), v), q
