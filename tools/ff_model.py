"""Row gathers a match step spends walking on from its landing row (CPU model; numpy).

A step from offset `off` of run k goes to LF(S[k]) + off, which lies in run LFrun[k] + t.  The compact row answers
t < 4 outright (cum_0..cum_3, spx_layout.h); past that the walk lands on run LFrun + 4 and moves on one row -- one
dependent gather -- at a time.  On the bench index (geometric run lengths, mean 8) that is 0.02 gathers per character.
It is unbounded in principle: a long run whose LF image covers many short runs (the move structure's known worst case;
Nishimoto & Tabei's balancing bounds it by splitting such runs).  This prints, for a few index shapes, how far the
shipped layout is from that bound and what a split rule "a piece ends where its image has crossed d run boundaries"
-- which the flatten step's pieces (spx_flatten.hip: runs of 2^16 and more) could take over -- would cost in extra rows.

    python tools/ff_model.py
"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spumoni_amd import synth


def spans(heads, lens):
    heads = heads.astype(np.int64).copy(); heads[heads <= 1] = 1
    r = heads.size
    ends = np.cumsum(lens); starts = ends - lens
    order = np.argsort(heads, kind="stable")
    lf = np.empty(r, np.int64); lf[order] = np.cumsum(lens[order]) - lens[order]
    a = np.searchsorted(starts, lf, side="right") - 1
    b = np.searchsorted(starts, lf + lens - 1, side="right") - 1
    return starts, lf, a, b


def report(name, heads, lens):
    lens = lens.astype(np.int64)
    r, n = heads.size, int(lens.sum())
    starts, lf, a, b = spans(heads, lens)
    span = b - a + 1
    # positions of run k whose step lands in run a[k] + t: the overlap of the image with that run; walking on costs
    # max(0, t - 3) gathers (t = 4 is the landing row itself when it is the right one... counted as 1: it is a gather
    # the layout did not foresee)
    extra = 0
    big = np.flatnonzero(span > 4)
    for k in big:
        dst = np.arange(a[k] + 4, b[k] + 1)
        lo = np.maximum(starts[dst], lf[k]); hi = np.minimum(starts[dst] + lens[dst], lf[k] + lens[k])
        extra += int(((hi - lo) * (dst - a[k] - 3)).sum())
    line = (f"{name}: r {r}, n {n} (n/r {n / r:.1f}), longest run {int(lens.max())}; image spans: max {int(span.max())} runs, "
            f"runs with span > 4: {big.size / r:.4f}; walking-on gathers per step, positions drawn uniformly: {extra / n:.3f}")
    print(line)
    for d in (8, 16, 64):
        more = int(np.maximum(0, (span - 1) // d).sum())
        print(f"    split at every {d:3d} crossed boundaries: + {more} rows ({100.0 * more / r:.2f} % of r), walking-on <= {max(0, d - 4)} gathers per step")


if __name__ == "__main__":
    raw = synth.statistical_rlbwt(1 << 20, 253, 8.0, seed=3, zipf=1.0)
    report("C3 statistical (sigma 253, mean run 8)", raw.heads.numpy(), raw.lens.numpy())
    raw = synth.statistical_rlbwt(1 << 20, 4, 60.0, seed=4, letters=b"ACGT")
    report("dna_m200 statistical (sigma 4, mean run 60)", raw.heads.numpy(), raw.lens.numpy())
    # the shape of tests/test_gpu_parity.py::test_long_runs_and_far_thresholds (1): 5 % of the runs 2^16..3*2^16 long among runs of 1..5
    rng = np.random.default_rng(5)
    rr = 4000
    idx = np.cumsum(rng.integers(1, 3, size=rr)) % 3
    lens = rng.integers(1, 6, size=rr).astype(np.int64)
    bigm = rng.random(rr) < 0.05
    lens[bigm] = rng.integers(1 << 16, 3 << 16, size=int(bigm.sum()))
    report("long runs among short ones (test shape)", np.frombuffer(b"ACG", dtype=np.uint8)[idx], lens)
    # heavy tail: Pareto run lengths (alpha 1.2) over a DNA alphabet
    rr = 1 << 18
    idx = np.cumsum(rng.integers(1, 4, size=rr)) % 4
    lens = np.minimum((rng.pareto(1.2, size=rr) + 1).astype(np.int64), 1 << 20)
    report("Pareto(1.2) run lengths, sigma 4", np.frombuffer(b"ACGT", dtype=np.uint8)[idx], lens)
    # a real BWT: 8 haplotypes of a 0.5 Mbp genome + reverse complements, undigested (spumoni run -n)
    base = synth.random_genome(500_000, seed=1)
    text, _ = synth.pangenome_text([base] + [synth.mutate(base, seed=s, snp=0.001, indel=0.0001) for s in range(2, 9)])
    raw = synth.index_from_text(torch.from_numpy(text), with_samples=False)
    report("real BWT, 8 haplotypes x 0.5 Mbp at 0.1 % divergence + reverse complements", raw.heads.numpy(), raw.lens.numpy())
