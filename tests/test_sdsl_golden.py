"""CPU: the serialised-index readers against byte strings DERIVED BY HAND from the serialize() functions
of simongog/sdsl-lite and maxrossi91/r-index -- not produced by tests/sdsl_files.py, whose writer follows
the same restatement as the reader and therefore proves self-consistency only.

Every constant below is worked out in the comment next to it.  What the reader INTERPRETS (sizes, widths,
packed words, the Elias-Fano low/high parts, the wavelet tree's bit vector and nodes) is pinned by these
strings.  The payloads of rank_support_v / select_support_mcl are skipped by the reader (it only needs
their framing to find the next field); they are written here as our reading of sdsl's construction and
marked as such.  None of this replaces a file written by an upstream build (tools/pin_upstream.py)."""
import os
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_BIN = os.path.join(ROOT, "spumoni_amd", "bin", "spumoni")


def u64(*v):
    return b"".join(struct.pack("<Q", x) for x in v)


def dump(tmp_path, kind, blob):
    p = tmp_path / (kind + ".bin")
    p.write_bytes(blob)
    r = subprocess.run([HOST_BIN, "dump-sdsl", kind, str(p)], capture_output=True, text=True)
    return r.returncode, r.stdout.strip(), r.stderr.strip()


def int_vector0(bits, width, *words):
    """sdsl::int_vector<0>::serialize: u64 size in BITS, u8 width, then ceil(bits / 64) 64-bit words."""
    return u64(bits) + bytes([width]) + u64(*words)


def test_int_vector_width_3(built_all, tmp_path):
    # [1, 5, 2] at 3 bits each: 9 bits; value i sits at bit 3 i of the word: 1 | 5 << 3 | 2 << 6 = 1 + 40 + 128 = 169 = 0xA9
    blob = bytes.fromhex("0900000000000000" "03" "A900000000000000")
    assert dump(tmp_path, "int_vector", blob) == (0, "1 5 2", "")


def test_int_vector_width_40_crosses_a_word(built_all, tmp_path):
    # [0x123456789A, 0xFFFFFFFFFF] at 40 bits (the width of SA samples of a text < 2^40): 80 bits, two words.
    # word 0 = first value | low 24 bits of the second << 40 = 0xFFFFFF_123456789A; word 1 = its high 16 bits = 0xFFFF
    blob = bytes.fromhex("5000000000000000" "28" "9A78563412FFFFFF" "FFFF000000000000")
    assert dump(tmp_path, "int_vector", blob) == (0, f"{0x123456789A} {0xFFFFFFFFFF}", "")


def test_int_vector_truncated_is_an_error(built_all, tmp_path):
    blob = bytes.fromhex("5000000000000000" "28" "9A78563412FFFFFF")  # second word missing
    rc, _, err = dump(tmp_path, "int_vector", blob)
    assert rc == 1 and "unexpected layout" in err


def test_bit_vector(built_all, tmp_path):
    # sdsl::bit_vector = int_vector<1>: u64 size in bits, words, NO width byte.  10 bits, ones at 0, 2, 9:
    # 1 + 4 + 512 = 517 = 0x0205
    blob = bytes.fromhex("0A00000000000000" "0502000000000000")
    assert dump(tmp_path, "bit_vector", blob) == (0, "1010000001", "")


def select_mcl(arg_cnt, logn, first_pos, span):
    """sdsl::select_support_mcl::serialize as we read it: u64 arg_cnt; if > 0: int_vector<0> superblock (one
    entry of logn bits per 4096 args: position of the superblock's first arg), bit_vector mini_or_long (EMPTY when
    no superblock is long), then per superblock one int_vector<0> miniblock of 64 entries of hi(span) + 1 bits.
    The reader only uses the framing."""
    if arg_cnt == 0:
        return u64(0)
    w = span.bit_length()  # bits::hi(span) + 1
    words = (64 * w + 63) // 64
    return u64(arg_cnt) + int_vector0(logn, logn, first_pos) + u64(0) + int_vector0(64 * w, w, *([0] * words))


def test_sparse_sd_vector_elias_fano(built_all, tmp_path):
    # ri::sparse_sd_vector::serialize: u64 u (length of the bit vector), u64 n (ones); if u > 0: sdsl::sd_vector<>.
    # The bit vector: 16 bits, ones at {3, 7, 12}.
    # sd_vector (sdsl sd_vector.hpp, constructor from a bit_vector): m = 3 ones, n = 16;
    #   logm = hi(3) + 1 = 2, logn = hi(16) + 1 = 5, logm != logn, so wl = logn - logm = 3 low bits per one;
    #   low[i]  = pos & 7            -> [3, 7, 4]   (int_vector<0>, width 3: 3 | 7 << 3 | 4 << 6 = 3 + 56 + 256 = 315 = 0x13B)
    #   high    = m + 2^logm = 7 bits, bit (pos >> 3) + i set -> bits 0 + 0, 0 + 1, 1 + 2 = {0, 1, 3} = 0b0001011 = 0x0B
    # serialize order: u64 size, u8 wl, low, high, select_support_mcl<1>(high), select_support_mcl<0>(high)
    low = int_vector0(9, 3, 0x13B)
    high = u64(7) + u64(0x0B)
    # selects over `high` (capacity 64 bits -> logn = hi(64) + 1 = 7): ones at 0, 1, 3 (first 0, span 3); zeros at 2, 4, 5, 6 (first 2, span 4)
    blob = u64(16, 3) + u64(16) + bytes([3]) + low + high + select_mcl(3, 7, 0, 3) + select_mcl(4, 7, 2, 4)
    assert dump(tmp_path, "sparse_sd", blob) == (0, "u 16 ones 3 7 12", "")
    # an empty vector is just the two counters
    assert dump(tmp_path, "sparse_sd", u64(0, 0)) == (0, "u 0 ones", "")
    # a vector whose high part names a position past the universe is refused
    bad = u64(16, 3) + u64(16) + bytes([3]) + low + u64(7) + u64(0b1100001) + select_mcl(3, 7, 0, 6) + select_mcl(4, 7, 1, 3)
    rc, _, err = dump(tmp_path, "sparse_sd", bad)  # ones decode to 3, (5-1)<<3|7 = 39, ... >= u
    assert rc == 1 and "unexpected layout" in err


def test_wt_huff_two_symbols(built_all, tmp_path):
    # ri::huff_string::serialize = sdsl::wt_huff<>::serialize (wt_pc.hpp): u64 size, u64 sigma, bit_vector bv,
    # rank_support_v<> (an int_vector<64>: u64 size in bits + words), select_support_mcl<1>, select_support_mcl<0>,
    # then the tree (_byte_tree::serialize): u64 #nodes, nodes {u64 bv_pos, u64 bv_pos_rank, u16 parent, u16 child[2]},
    # u16 c_to_leaf[256], u64 path[256] (code in the low bits, its length << 56).
    # Sequence "ABBA" (65 66 66 65): two symbols of equal frequency -> codes A = 0, B = 1 (one bit each); the root
    # (node 0) owns bv[0, 4) = 0 1 1 0 -> 0b0110 = 6; leaves are nodes 1 (A) and 2 (B).
    bv = u64(4) + u64(0b0110)
    rank_v = u64(128) + u64(0, 0)  # ((64 >> 9) + 1) << 1 = 2 words; payload not interpreted
    undef = 0xFFFF

    def node(bv_pos, rank, parent, c0, c1):
        return struct.pack("<QQHHH", bv_pos, rank, parent, c0, c1)

    nodes = node(0, 0, undef, 1, 2) + node(65, 0, 0, undef, undef) + node(66, 0, 0, undef, undef)
    c_to_leaf = [undef] * 256
    c_to_leaf[65], c_to_leaf[66] = 1, 2
    path = [0] * 256
    path[65], path[66] = (1 << 56) | 0, (1 << 56) | 1
    blob = (u64(4, 2) + bv + rank_v + select_mcl(2, 7, 1, 1) + select_mcl(2, 7, 0, 3) + u64(3) + nodes +
            struct.pack("<256H", *c_to_leaf) + struct.pack("<256Q", *path))
    assert dump(tmp_path, "wt_huff", blob) == (0, "65 66 66 65", "")


def test_wt_huff_three_symbols_two_levels(built_all, tmp_path):
    # "CACBCA C" = C A C B C A C: freq C 4, A 2, B 1.  Huffman: merge B + A first (3), then with C (4):
    #   root (node 0): C -> 1, {A, B} -> 0;  inner node 1 (the 0 child): A -> 1?  Any assignment works as long as the
    #   stream is consistent: take node 1: B -> 0, A -> 1.  Leaves: node 2 = C, node 3 = B, node 4 = A.
    # bv is the concatenation of the nodes' bit vectors in node order:
    #   root  bits (one per character)      C A C B C A C -> 1 0 1 0 1 0 1      at bv_pos 0, 7 bits
    #   node1 bits (one per A / B, in order) A B A        -> 1 0 1              at bv_pos 7, 3 bits
    #   10 bits in all: bit i set for i in {0, 2, 4, 6, 7, 9} = 1 + 4 + 16 + 64 + 128 + 512 = 725 = 0x2D5
    bv = u64(10) + u64(0x2D5)
    rank_v = u64(128) + u64(0, 0)
    undef = 0xFFFF

    def node(bv_pos, rank, parent, c0, c1):
        return struct.pack("<QQHHH", bv_pos, rank, parent, c0, c1)

    # bv_pos_rank = ones in bv before bv_pos: root 0; node 1 starts at 7 with ones {0,2,4,6} before it = 4
    nodes = (node(0, 0, undef, 1, 2) + node(7, 4, 0, 3, 4) + node(67, 0, 0, undef, undef) +
             node(66, 0, 1, undef, undef) + node(65, 0, 1, undef, undef))
    c_to_leaf = [undef] * 256
    c_to_leaf[67], c_to_leaf[66], c_to_leaf[65] = 2, 3, 4
    path = [0] * 256
    blob = (u64(7, 3) + bv + rank_v + select_mcl(6, 7, 0, 9) + select_mcl(4, 7, 1, 7) + u64(5) + nodes +
            struct.pack("<256H", *c_to_leaf) + struct.pack("<256Q", *path))
    assert dump(tmp_path, "wt_huff", blob) == (0, "67 65 67 66 67 65 67", "")
