"""Brute-force ground truth for tiny texts (first principles, no oracle / product code)."""
import numpy as np


def naive_sa(t):
    """t: list of ints incl. terminator (unique smallest, last). Returns suffix array."""
    n = len(t)
    return sorted(range(n), key=lambda i: t[i:])


def naive_bwt(t):
    sa = naive_sa(t)
    n = len(t)
    return sa, [t[(s - 1) % n] for s in sa]


def naive_lcp(t, sa):
    n = len(t)
    lcp = [0] * n
    for i in range(1, n):
        a, b = sa[i - 1], sa[i]
        l = 0
        while a + l < n and b + l < n and t[a + l] == t[b + l]:
            l += 1
        lcp[i] = l
    return lcp


def runs_of(bwt):
    heads, lens = [], []
    for c in bwt:
        if heads and heads[-1] == c:
            lens[-1] += 1
        else:
            heads.append(c)
            lens.append(1)
    return heads, lens


def true_ms(text, read):
    """Matching statistics: ms[i] = longest prefix of read[i:] that occurs in text."""
    tb = bytes(text)
    rb = bytes(read)
    m = len(rb)
    out = []
    for i in range(m):
        l = 0
        while i + l < m and tb.find(rb[i : i + l + 1]) >= 0:
            l += 1
        out.append(l)
    return out


def rank_brute(bwt, p, c):
    return sum(1 for x in bwt[:p] if x == c)


def select_brute(bwt, i, c):
    cnt = 0
    for p, x in enumerate(bwt):
        if x == c:
            if cnt == i:
                return p
            cnt += 1
    raise IndexError


# ---------------------------------------------------------------------------------------------
# minimizer digestion, written as a specification (slices and min(), no queue): what
# oracle/orc_digest.c and the HIP kernel must both produce.  See DESIGN.md 4.4 for the
# assumptions about bonsai this encodes.
LEX_XOR_MASK = 0xE37E28C4271B5A2D


def digest_spec(kind, k, w, seq: bytes, charhash):
    """kind 1: -m (promoted), kind 2: -a (DNA letters).  charhash: T[A], T[C], T[G], T[T]."""
    code = {65: 0, 67: 1, 71: 2, 84: 3}
    wsz = max(1, w - k + 1)

    def rotl8(x, s):
        s %= 8
        return ((x << s) | (x >> (8 - s))) & 0xFF if s else x

    stream = []  # (score, element) of every k-mer made of ACGT only, in read order
    for i in range(k - 1, len(seq)):
        win = seq[i - k + 1 : i + 1]
        if any(c not in code for c in win):
            continue
        if kind == 2:
            km = 0
            for c in win:
                km = km * 4 + code[c]
            stream.append((km ^ LEX_XOR_MASK, km))
        else:
            h = 0
            for j, c in enumerate(win):
                h ^= rotl8(charhash[code[c]], k - 1 - j)
            stream.append((h, h))
    reports = [min(stream[t - wsz + 1 : t + 1])[1] for t in range(wsz - 1, len(stream))]
    out = bytearray()
    last = None
    for x in reports:
        if last is None or (last & 0xFF) != x:  # mseq_vec is a vector<uint8_t>
            last = x
            if kind == 2:
                out += bytes("ACGT"[(x >> (2 * (k - 1 - j))) & 3].encode()[0] for j in range(k))
            else:
                out.append(x if x > 2 else x + 3)
    return bytes(out)
