"""Synthetic index / read generators (ours; no reference code).

The reference ships no data and its index builder shells out to PFP binaries
that are not available offline, so tests and bench make
their own inputs:

* ``index_from_text``  -- a REAL run-length BWT of a text (suffix array by
  prefix doubling, LCP from the doubling ranks, thresholds = first arg-min LCP
  between consecutive same-letter runs, SA samples, document ids), emitted as
  the raw per-run arrays of SURVEY Appendix A.1 (`.bwt.heads/.bwt.len/
  .thr_pos/.ssa/.esa`).
* ``statistical_rlbwt`` -- a statistical RLBWT (random heads / geometric run
  lengths / thresholds in the legal interval) for human-pangenome scale, where
  building a true BWT offline is not possible.  It is a valid input for the
  query algorithm (every rank/select it can reach is defined) but is not the
  BWT of any text.

Everything is vectorised with torch so the same code runs on the CPU (tests
here) and on an MI355X (bench / -m gpu tests).
"""
from __future__ import annotations

import dataclasses
from typing import Optional, Sequence

import numpy as np
import torch

TERMINATOR = 1  # include/ms_rle_string.hpp:21 (heads 0/1 both mean terminator)


@dataclasses.dataclass
class RawIndex:
    """Raw per-run arrays (what newscanNT.x + pfp_thresholds -r write)."""

    heads: torch.Tensor  # u8  [r]
    lens: torch.Tensor  # i64 [r]
    thr: torch.Tensor  # i64 [r]   raw .thr_pos values (0 for first run of a letter)
    n: int
    ssa: Optional[torch.Tensor] = None  # i64 [r] stored samples (right ? right-1 : n-1)
    esa: Optional[torch.Tensor] = None
    doc_start: Optional[torch.Tensor] = None  # i64 [r]
    doc_end: Optional[torch.Tensor] = None
    text: Optional[torch.Tensor] = None  # u8 [n-1] text without terminator (MS extension)

    @property
    def r(self) -> int:
        return int(self.heads.numel())

    def cpu(self) -> "RawIndex":
        kw = {}
        for f in dataclasses.fields(self):
            v = getattr(self, f.name)
            kw[f.name] = v.cpu() if isinstance(v, torch.Tensor) else v
        return RawIndex(**kw)

    def write_raw_files(self, prefix: str) -> None:
        """Write `<prefix>.bwt.heads/.bwt.len/.thr_pos[/.ssa/.esa]` (5-byte LE)."""

        def five(t: torch.Tensor) -> bytes:
            a = t.cpu().numpy().astype("<u8")
            return a.view(np.uint8).reshape(-1, 8)[:, :5].tobytes()

        with open(prefix + ".bwt.heads", "wb") as f:
            f.write(self.heads.cpu().numpy().tobytes())
        with open(prefix + ".bwt.len", "wb") as f:
            f.write(five(self.lens))
        with open(prefix + ".thr_pos", "wb") as f:
            f.write(five(self.thr))
        if self.ssa is not None:
            starts = torch.cumsum(self.lens, 0) - self.lens
            for name, samp, left in (("ssa", self.ssa, starts), ("esa", self.esa, starts + self.lens - 1)):
                # stored = right ? right-1 : n-1   =>   right = (stored + 1) % n
                right = (samp + 1) % self.n
                pair = torch.stack([left, right], 1).reshape(-1)
                with open(prefix + "." + name, "wb") as f:
                    f.write(five(pair))


# --------------------------------------------------------------------------
# suffix array + LCP by prefix doubling
# --------------------------------------------------------------------------
def suffix_array_lcp(text: torch.Tensor):
    """text: u8/i64 [n], last symbol must be the unique smallest (terminator).

    Returns (sa i64[n], lcp i64[n]) with lcp[i] = LCP(suffix sa[i-1], suffix sa[i]), lcp[0]=0.
    """
    dev = text.device
    n = int(text.numel())
    t = text.to(torch.int64)
    # rank by first character (dense)
    uniq, rank = torch.unique(t, return_inverse=True)
    levels = [rank.to(torch.int32)]
    k = 1
    ar = torch.arange(n, device=dev)
    while True:
        nr = int(rank.max().item()) + 1
        if nr == n:
            break
        second = torch.zeros(n, dtype=torch.int64, device=dev)
        if k < n:
            second[: n - k] = rank[k:] + 1  # 0 = past the end
        key = rank * (nr + 1) + second
        skey, order = torch.sort(key)
        newr = torch.zeros(n, dtype=torch.int64, device=dev)
        newr[1:] = torch.cumsum((skey[1:] != skey[:-1]).to(torch.int64), 0)
        rank = torch.empty_like(newr)
        rank[order] = newr
        levels.append(rank.to(torch.int32))
        k *= 2
        del second, key, skey, order, newr
    sa = torch.empty(n, dtype=torch.int64, device=dev)
    sa[rank] = ar
    # LCP of neighbours from the stored doubling levels, top-down
    a = sa[:-1].clone()
    b = sa[1:].clone()
    l = torch.zeros(n - 1, dtype=torch.int64, device=dev)
    for lev in range(len(levels) - 1, -1, -1):
        kk = 1 << lev
        ra = levels[lev][torch.clamp(a + l, max=n - 1)]
        rb = levels[lev][torch.clamp(b + l, max=n - 1)]
        ok = (ra == rb) & (a + l < n) & (b + l < n)
        l = l + ok.to(torch.int64) * kk
    lcp = torch.zeros(n, dtype=torch.int64, device=dev)
    lcp[1:] = l
    return sa, lcp


def index_from_text(
    text: torch.Tensor,
    doc_lengths: Optional[Sequence[int]] = None,
    with_samples: bool = True,
) -> RawIndex:
    """Real RLBWT + thresholds + SA samples of `text` (u8, every byte >= 2).

    A terminator (smallest symbol) is appended; it shows up as a head byte 0 in
    `heads` exactly like Big-BWT writes it (the loader rewrites it to 1).
    doc_lengths: lengths of the concatenated documents (sum == len(text)).
    """
    dev = text.device
    assert int(text.min().item()) >= 2, "text bytes must be >= 2 (0/1 are the terminator)"
    n = int(text.numel()) + 1
    t = torch.cat([text.to(torch.int64), torch.zeros(1, dtype=torch.int64, device=dev)])
    sa, lcp = suffix_array_lcp(t)
    bwt = t[(sa - 1) % n]  # BWT[i] = T[SA[i]-1]
    # runs
    is_start = torch.ones(n, dtype=torch.bool, device=dev)
    is_start[1:] = bwt[1:] != bwt[:-1]
    starts = torch.nonzero(is_start).reshape(-1)
    r = int(starts.numel())
    ends = torch.cat([starts[1:] - 1, torch.tensor([n - 1], device=dev)])
    heads = bwt[starts].to(torch.uint8)
    lens = ends - starts + 1
    # thresholds: for each run k that is not the first of its letter, the first
    # arg-min of LCP over (end of previous same-letter run, start of run k].
    thr = torch.zeros(r, dtype=torch.int64, device=dev)
    key = lcp * (n + 1) + torch.arange(n, device=dev)  # min key -> first arg-min
    big = torch.iinfo(torch.int64).max
    heads64 = heads.to(torch.int64)
    for c in torch.unique(heads64).tolist():
        ks = torch.nonzero(heads64 == c).reshape(-1)
        if ks.numel() < 2:
            continue
        lo = ends[ks[:-1]] + 1  # inclusive
        hi = starts[ks[1:]]  # inclusive
        thr[ks[1:]] = _range_min(key, lo, hi, big) % (n + 1)
    raw = RawIndex(heads=heads, lens=lens, thr=thr, n=n, text=text.to(torch.uint8))
    if with_samples:
        # `.ssa/.esa` right value = SA at the run's first/last position;
        # stored sample = right ? right-1 : n-1 (compute_ms_pml.cpp:433)
        raw.ssa = (sa[starts] - 1) % n
        raw.esa = (sa[ends] - 1) % n
        if doc_lengths is not None:
            # doc id = number of document ends <= text position of the BWT char
            # (src/doc_array.cpp:31-91; last document absorbs the terminator)
            end_pos = torch.cumsum(torch.tensor(list(doc_lengths), dtype=torch.int64, device=dev), 0)
            end_pos[-1] += 1
            raw.doc_start = torch.searchsorted(end_pos, raw.ssa, right=True)
            raw.doc_end = torch.searchsorted(end_pos, raw.esa, right=True)
    return raw


def _range_min(key: torch.Tensor, lo: torch.Tensor, hi: torch.Tensor, big: int) -> torch.Tensor:
    """min(key[lo[i]..hi[i]]) for disjoint ascending ranges, vectorised (segment reduce)."""
    n = key.numel()
    m = lo.numel()
    seg = torch.full((n,), -1, dtype=torch.int64, device=key.device)
    # mark positions: +1 at lo, -1 after hi  -> inside ranges cumsum == 1
    mark = torch.zeros(n + 1, dtype=torch.int64, device=key.device)
    mark.index_add_(0, lo, torch.ones_like(lo))
    mark.index_add_(0, hi + 1, -torch.ones_like(hi))
    inside = torch.cumsum(mark[:n], 0) > 0
    sid = torch.zeros(n + 1, dtype=torch.int64, device=key.device)
    sid.index_add_(0, lo, torch.ones_like(lo))
    sid = torch.cumsum(sid[:n], 0) - 1
    seg = torch.where(inside, sid, torch.full_like(sid, m))
    out = torch.full((m + 1,), big, dtype=torch.int64, device=key.device)
    out.scatter_reduce_(0, seg, key, reduce="amin", include_self=True)
    return out[:m]


# --------------------------------------------------------------------------
# statistical RLBWT (human-pangenome scale stand-in)
# --------------------------------------------------------------------------
def statistical_rlbwt(
    r: int,
    sigma: int,
    mean_run: float,
    seed: int,
    device: torch.device | str = "cpu",
    zipf: float = 0.0,
    letters: Optional[Sequence[int]] = None,
    with_samples: bool = False,
    n_docs: int = 0,
) -> RawIndex:
    """Random run heads (no equal neighbours), geometric run lengths, thresholds
    uniform in (end of previous same-letter run, start of this run].

    letters: the byte values to use (default 3..3+sigma-1 => promoted-minimizer
    style alphabet incl. bytes >= 128; pass b"ACGT" for DNA).  Run 0 is the
    terminator run (head 0, length 1) so the index has the same shape as a real
    one.  zipf>0 draws heads from a Zipf(zipf) law over the letters.
    """
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    if letters is None:
        letters = list(range(3, 3 + sigma))
    lt = torch.tensor(list(letters), dtype=torch.int64, device=dev)
    sigma = int(lt.numel())
    assert sigma >= 2
    # heads: draw i.i.d., then fix equal neighbours by shifting to the next letter
    if zipf > 0:
        w = 1.0 / torch.arange(1, sigma + 1, dtype=torch.float64, device=dev) ** zipf
        cdf = torch.cumsum(w / w.sum(), 0)
        idx = torch.empty(r, dtype=torch.int64, device=dev)
        CH = 1 << 26
        for s in range(0, r, CH):
            e = min(r, s + CH)
            u = torch.rand(e - s, generator=g, device=dev, dtype=torch.float64)
            idx[s:e] = torch.clamp(torch.searchsorted(cdf, u), max=sigma - 1)
    else:
        idx = torch.randint(0, sigma, (r,), generator=g, device=dev)
    # remove equal neighbours: idx'[i] = (idx[i] + d[i]) with d chosen so that
    # consecutive values differ: use the "differences" trick -- draw steps in
    # [1, sigma-1] wherever a collision exists and re-scan (few passes suffice)
    for _ in range(64):
        eq = torch.zeros(r, dtype=torch.bool, device=dev)
        eq[1:] = idx[1:] == idx[:-1]
        # only fix odd/even alternately to avoid creating new collisions in lock-step
        if not bool(eq.any()):
            break
        bump = torch.randint(1, sigma, (r,), generator=g, device=dev)
        idx = torch.where(eq, (idx + bump) % sigma, idx)
    else:
        raise RuntimeError("could not de-collide heads")
    heads = lt[idx].to(torch.uint8)
    # geometric lengths with the requested mean (>= 1)
    if mean_run <= 1.0:
        lens = torch.ones(r, dtype=torch.int64, device=dev)
    else:
        p = 1.0 / mean_run
        u = torch.rand(r, generator=g, device=dev, dtype=torch.float64)
        lens = (torch.floor(torch.log1p(-u) / np.log1p(-p)).to(torch.int64) + 1).clamp_(min=1)
    # terminator run
    heads[0] = 0
    lens[0] = 1
    if r > 1 and int(heads[1]) == 0:
        heads[1] = int(lt[0])
    return raw_from_runs(heads, lens, g, with_samples, n_docs)


def raw_from_runs(heads: torch.Tensor, lens: torch.Tensor, seed_or_gen, with_samples: bool = False,
                  n_docs: int = 0) -> RawIndex:
    """RawIndex over given run heads (u8, no equal neighbours, exactly one terminator run with head
    0) and run lengths: thresholds uniform in (end of previous same-letter run, start of this run],
    random SA samples / document ids."""
    dev = heads.device
    r = int(heads.numel())
    if isinstance(seed_or_gen, torch.Generator):
        g = seed_or_gen
    else:
        g = torch.Generator(device=dev)
        g.manual_seed(int(seed_or_gen))
    ends_excl = torch.cumsum(lens, 0)
    starts = ends_excl - lens
    n = int(ends_excl[-1].item())
    # thresholds: per letter, uniform in (end of prev c-run, start of this c-run]
    # = prev_end_excl .. start  (inclusive both) where prev_end_excl = prev start+len
    h64 = heads.to(torch.int64)
    order = torch.sort(h64, stable=True).indices  # runs grouped by letter, ascending index
    hs = h64[order]
    first = torch.ones(r, dtype=torch.bool, device=dev)
    first[1:] = hs[1:] != hs[:-1]
    prev_end = torch.zeros(r, dtype=torch.int64, device=dev)
    prev_end[1:] = ends_excl[order[:-1]]
    lo = prev_end  # first legal value (position right after previous c-run)
    hi = starts[order]
    u = torch.rand(r, generator=g, device=dev, dtype=torch.float64)
    t = lo + torch.floor(u * (hi - lo + 1).to(torch.float64)).to(torch.int64)
    t = torch.minimum(torch.maximum(t, lo), hi)
    t = torch.where(first, torch.zeros_like(t), t)
    thr = torch.empty(r, dtype=torch.int64, device=dev)
    thr[order] = t
    raw = RawIndex(heads=heads, lens=lens, thr=thr, n=n)
    if with_samples:
        raw.ssa = torch.randint(0, n, (r,), generator=g, device=dev)
        raw.esa = torch.randint(0, n, (r,), generator=g, device=dev)
    if n_docs > 0:
        raw.doc_start = torch.randint(0, n_docs, (r,), generator=g, device=dev)
        raw.doc_end = torch.randint(0, n_docs, (r,), generator=g, device=dev)
    return raw


# --------------------------------------------------------------------------
# texts and reads
# --------------------------------------------------------------------------
_DNA = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTN", b"TGCAN"):
    _COMP[_a] = _b


def random_genome(length: int, seed: int) -> np.ndarray:
    rng = np.random.default_rng(seed)
    return _DNA[rng.integers(0, 4, size=length)]


def revcomp(seq: np.ndarray) -> np.ndarray:
    return _COMP[seq[::-1]]


def mutate(genome: np.ndarray, seed: int, snp: float = 0.01, indel: float = 0.001) -> np.ndarray:
    """Derivative strain: SNPs + short (1-50 bp) indels."""
    rng = np.random.default_rng(seed)
    g = genome.copy()
    pos = np.nonzero(rng.random(g.size) < snp)[0]
    g[pos] = _DNA[(np.searchsorted(_DNA, g[pos]) + rng.integers(1, 4, size=pos.size)) % 4]
    ev = np.nonzero(rng.random(g.size) < indel)[0]
    pieces, last = [], 0
    for p in ev.tolist():
        pieces.append(g[last:p])
        ln = int(rng.integers(1, 51))
        if rng.random() < 0.5:
            pieces.append(_DNA[rng.integers(0, 4, size=ln)])  # insertion
            last = p
        else:
            last = min(g.size, p + ln)  # deletion
    pieces.append(g[last:])
    return np.concatenate(pieces)


def pangenome_text(genomes: Sequence[np.ndarray], add_revcomp: bool = True):
    """Concatenate genomes (+ reverse complements); returns (text u8, doc_lengths)."""
    parts, doc_lengths = [], []
    for gnm in genomes:
        seqs = [gnm, revcomp(gnm)] if add_revcomp else [gnm]
        doc_lengths.append(sum(s.size for s in seqs))
        parts.extend(seqs)
    return np.concatenate(parts), doc_lengths


def sample_reads(text: np.ndarray, nreads: int, length: int, seed: int, err: float = 0.01,
                 null_fraction: float = 0.5):
    """50% reads sampled from the text with substitution errors, 50% 'null' reads =
    such a sample reversed (the reference's own null model, compute_ms_pml.cpp:1464-1465).

    Returns (seqs u8 [nreads*length], offsets i64 [nreads+1]) -- fixed length reads.
    """
    rng = np.random.default_rng(seed)
    start = rng.integers(0, text.size - length, size=nreads)
    idx = start[:, None] + np.arange(length)[None, :]
    reads = text[idx]
    if err > 0:
        e = rng.random(reads.shape) < err
        sub = _DNA[rng.integers(0, 4, size=int(e.sum()))]
        reads[e] = sub
    null = rng.random(nreads) < null_fraction
    reads[null] = reads[null, ::-1]
    offs = np.arange(nreads + 1, dtype=np.int64) * length
    return np.ascontiguousarray(reads.reshape(-1)), offs


def simulate_reads(raw: RawIndex, nreads: int, length: int, seed: int,
                   positive_fraction: float = 0.5, f_mis: float = 0.02, warmup: int = 0):
    """Reads for a statistical RLBWT (no text exists to sample from).

    "Positive" reads are produced by *simulating the backward search itself*
    (a vectorised torch walk over all reads at once): at every step the read's
    next character is the head of the run the walk currently sits in (so the
    search will match) except with probability `f_mis`, where a different
    random letter is drawn and the walk follows the reference's threshold jump
    (compute_ms_pml.cpp:251-278).  The remaining reads are uniform over the
    letters.  Requires thresholds that satisfy thr[k] <= start[k] (true for
    `statistical_rlbwt` and for real indexes).

    warmup: every search starts at pos = n - 1 (compute_ms_pml.cpp:243), so positive reads
    simulated from there all spell the SAME path until their first mismatch (with f_mis = 0.02
    and 44 characters, 41 % of them are the same read, and the walks of the others share their
    first ~50 gathers: cache hits that no real batch of reads would see).  The first `warmup`
    searched characters (the read's right end) are therefore drawn at random -- each sends the
    walk to another letter's runs -- so that after them the walks of different reads are spread
    over the whole index (sigma^warmup places) before the matching stretch begins.

    Runs on raw.heads.device.  Returns (seqs u8 [nreads*length], offsets i64).
    """
    dev = raw.heads.device
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    r, n = raw.r, raw.n
    h64 = raw.heads.to(torch.int64).clone()
    h64[h64 <= TERMINATOR] = TERMINATOR
    ends_excl = torch.cumsum(raw.lens, 0)
    starts = ends_excl - raw.lens
    order = torch.sort(h64, stable=True).indices          # runs grouped by letter
    hs = h64[order]
    comp = hs * r + order                                  # ascending composite key
    lf_sorted = torch.cumsum(raw.lens[order], 0) - raw.lens[order]  # LF(start of run order[i])
    lf_of_run = torch.empty(r, dtype=torch.int64, device=dev)
    lf_of_run[order] = lf_sorted
    seg_lo = torch.searchsorted(hs, torch.arange(256, device=dev))
    seg_hi = torch.searchsorted(hs, torch.arange(256, device=dev), right=True)
    letters = torch.unique(h64)
    letters = letters[letters > TERMINATOR]
    nl = int(letters.numel())
    npos = int(round(nreads * positive_fraction))
    seqs = letters[torch.randint(0, nl, (nreads, length), generator=g, device=dev)].to(torch.uint8)
    if npos > 0:
        out = torch.empty((npos, length), dtype=torch.uint8, device=dev)
        pos = torch.full((npos,), n - 1, dtype=torch.int64, device=dev)
        for i in range(length):
            at_end = pos >= n
            k = torch.clamp(torch.searchsorted(starts, pos, right=True) - 1, max=r - 1)
            head = h64[k]
            rnd = letters[torch.randint(0, nl, (npos,), generator=g, device=dev)]
            mis = (torch.rand(npos, generator=g, device=dev) < f_mis) | (head <= TERMINATOR) | at_end
            if i < warmup:
                mis = torch.ones_like(mis)
            c = torch.where(mis, rnd, head)
            out[:, length - 1 - i] = c.to(torch.uint8)
            stay = (c == head) & ~at_end
            # jump: number of c-runs before run k (k = r when pos == n)
            kk = torch.where(at_end, torch.full_like(k, r), k)
            jg = torch.searchsorted(comp, c * r + kk)
            has_succ = jg < seg_hi[c]
            qs = order[torch.clamp(jg, max=r - 1)]
            thr = torch.where(has_succ, raw.thr[qs], torch.full_like(pos, n + 1))
            # C14: first run of a letter has threshold 0
            thr = torch.where(has_succ & (jg == seg_lo[c]), torch.zeros_like(thr), thr)
            use_pred = pos < thr
            qp = order[torch.clamp(jg - 1, min=0)]
            q = torch.where(use_pred, qp, qs)
            jpos = torch.where(use_pred, starts[qp] + raw.lens[qp] - 1, starts[qs])
            newpos = torch.where(stay, pos, jpos)
            run = torch.where(stay, k, q)
            pos = lf_of_run[run] + (newpos - starts[run])
        seqs[:npos] = out
        perm = torch.randperm(nreads, generator=g, device=dev)
        seqs = seqs[perm]
    offs = torch.arange(nreads + 1, dtype=torch.int64, device=dev) * length
    return seqs.reshape(-1).contiguous(), offs
