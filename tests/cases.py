"""Shared builders of (index, reads) parity cases (used by CPU and GPU tests)."""
import numpy as np
import torch

from spumoni_amd import synth


def reads_mixed(rng, text, letters, nreads, maxlen, extra_letters=()):
    """ragged reads: substrings with errors, random, with absent letters; some empty."""
    reads = []
    pool = list(letters) + list(extra_letters)
    for q in range(nreads):
        m = int(rng.integers(0, maxlen)) if q % 17 else 0  # every 17th read is empty
        kind = rng.integers(0, 3)
        if m == 0:
            rd = np.zeros(0, dtype=np.uint8)
        elif kind == 0 and text is not None and text.size > m:
            s = int(rng.integers(0, text.size - m))
            rd = text[s : s + m].copy()
            for _ in range(int(rng.integers(0, 4))):
                rd[rng.integers(0, m)] = letters[rng.integers(0, len(letters))]
        elif kind == 1:
            rd = np.asarray(letters, dtype=np.uint8)[rng.integers(0, len(letters), size=m)]
        else:
            rd = np.asarray(pool, dtype=np.uint8)[rng.integers(0, len(pool), size=m)]
        reads.append(rd.astype(np.uint8))
    offs = np.concatenate([[0], np.cumsum([r.size for r in reads])]).astype(np.int64)
    seqs = np.concatenate(reads) if reads else np.zeros(0, dtype=np.uint8)
    return seqs, offs


def repetitive_text(rng, n, letters):
    letters = np.asarray(letters, dtype=np.uint8)
    base = letters[rng.integers(0, letters.size, size=max(8, n // 5))]
    parts, tot = [], 0
    while tot < n:
        p = base.copy()
        for _ in range(int(rng.integers(0, 6))):
            p[rng.integers(0, p.size)] = letters[rng.integers(0, letters.size)]
        p = p[int(rng.integers(0, p.size // 2)) :]
        parts.append(p)
        tot += p.size
    return np.concatenate(parts)[:n]


def real_case(seed, n, letters, ndocs=3, device="cpu"):
    rng = np.random.default_rng(seed)
    text = repetitive_text(rng, n, letters)
    cuts = sorted(rng.choice(np.arange(1, n), size=ndocs - 1, replace=False).tolist())
    doc_lengths = np.diff([0] + cuts + [n]).tolist()
    raw = synth.index_from_text(torch.from_numpy(text).to(device), doc_lengths=doc_lengths)
    return raw, text
