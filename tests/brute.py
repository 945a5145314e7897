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
