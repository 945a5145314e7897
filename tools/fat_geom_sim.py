"""Fat-table geometries on a REAL digested BWT (CPU model; numpy).  Round 6.

Builds the bench's real-BWT leg at a CPU-sized genome (10 haplotypes + reverse complements, digested -m k=4 w=11),
replays the PML walk (compute_ms_pml.cpp:238-286) over digested reads (half from the text with 1 % substitutions, half
reversed) and records every threshold jump as (letter c, run k).  Then prices candidate geometries of the fat table
(spx_layout.h) at the same number of slots: how many gathers a jump costs.

    python tools/fat_geom_sim.py [genome_bp] [slots_per_run] [nreads]

The jump list is cached in /tmp (fat_geom_<genome_bp>.npz).
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spumoni_amd import synth  # noqa: E402

genome_bp = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
spr = float(sys.argv[2]) if len(sys.argv) > 2 else 6.8
nreads = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
cache = f"/tmp/fat_geom_{genome_bp}_{nreads}.npz"


def build():
    import oracle  # (tool: the CPU digestion)

    t0 = time.time()
    base = synth.random_genome(genome_bp, seed=1)
    genomes = [base] + [synth.mutate(base, seed=sd) for sd in range(2, 11)]
    k, w = 4, 11
    parts, dna = [], []
    for g in genomes:
        for seq in (g, synth.revcomp(g)):
            d, _ = oracle.digest_batch(oracle.DIGEST_PROMOTED, k, w, seq, np.array([0, seq.size], dtype=np.uint64))
            parts.append(d.copy())
            dna.append(seq)
    dtext = np.concatenate(parts)
    text = np.concatenate(dna)
    print(f"digested text {dtext.size} chars ({time.time() - t0:.1f}s)", flush=True)
    raw = synth.index_from_text(torch.from_numpy(dtext), with_samples=False)
    print(f"index n={raw.n} r={raw.r} ({time.time() - t0:.1f}s)", flush=True)
    rng = np.random.default_rng(12)
    bp = 200
    start = rng.integers(0, text.size - bp, nreads)
    reads = text[start[:, None] + np.arange(bp)[None, :]]
    sub = rng.random((nreads, bp)) < 0.01
    reads = np.where(sub, np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (nreads, bp))], reads)
    null = rng.random(nreads) < 0.5
    reads = np.where(null[:, None], reads[:, ::-1], reads).astype(np.uint8)
    ds, do = oracle.digest_batch(oracle.DIGEST_PROMOTED, k, w, reads.reshape(-1).copy(), np.arange(nreads + 1, dtype=np.uint64) * bp)
    heads = raw.heads.numpy().astype(np.int64)
    heads[heads <= 1] = 1
    lens = raw.lens.numpy().astype(np.int64)
    thr = raw.thr.numpy().astype(np.int64)
    return heads, lens, thr, ds, do.astype(np.int64)


def walk(heads, lens, thr, ds, do):
    """the PML walk, all reads in step; returns the jumps as (c, k, pred) and the number of steps"""
    r = heads.size
    ends = np.cumsum(lens)
    starts = ends - lens
    n = int(ends[-1])
    order = np.argsort(heads, kind="stable")
    lf_sorted = np.cumsum(lens[order]) - lens[order]
    lf_start = np.empty(r, np.int64)
    lf_start[order] = lf_sorted
    hs = heads[order]
    comp = hs * r + order
    seg_lo = np.searchsorted(hs, np.arange(257))
    seg_hi = seg_lo[1:]
    seg_lo = seg_lo[:-1]
    nr = do.size - 1
    mlen = do[1:] - do[:-1]
    pos = np.full(nr, n - 1, np.int64)
    jc, jk = [], []
    steps = 0
    for i in range(int(mlen.max())):
        live = mlen > i
        idx = np.nonzero(live)[0]
        c = ds[do[idx + 1] - 1 - i].astype(np.int64)
        p = pos[idx]
        k = np.minimum(np.searchsorted(starts, p, side="right") - 1, r - 1)
        present = seg_hi[c] > seg_lo[c]
        match = (heads[k] == c) & (c < 128) & present
        quirk = (heads[k] == c) & (c >= 128) & present
        jg = np.searchsorted(comp, c * r + k)
        has_succ = jg < seg_hi[c]
        qs = order[np.minimum(jg, r - 1)]
        th = np.where(has_succ, thr[qs], n + 1)
        th = np.where(has_succ & (jg == seg_lo[c]), 0, th)
        use_pred = p < th
        qp = order[np.maximum(jg - 1, 0)]
        q = np.where(use_pred, qp, qs)
        jpos = np.where(use_pred, starts[qp] + lens[qp] - 1, starts[qs])
        stay = match | quirk
        run = np.where(stay, k, q)
        newpos = np.where(stay, p, jpos)
        npos = lf_start[run] + (newpos - starts[run])
        # absent letters: LF = F[c] (rare: ignore the exact landing, any position does for the statistics)
        npos = np.where(present, npos, 0)
        jm = present & ~stay
        jc.append(c[jm])
        jk.append(k[jm])
        steps += idx.size
        pos[idx] = npos
    return np.concatenate(jc), np.concatenate(jk), steps, order, seg_lo, seg_hi


if os.path.exists(cache):
    z = np.load(cache)
    heads, jc, jk, steps = z["heads"], z["jc"], z["jk"], int(z["steps"])
else:
    heads, lens, thr, ds, do = build()
    t0 = time.time()
    jc, jk, steps, *_ = walk(heads, lens, thr, ds, do)
    print(f"walk {steps} steps, {jc.size} jumps ({time.time() - t0:.1f}s)", flush=True)
    np.savez(cache, heads=heads, jc=jc, jk=jk, steps=steps)

r = heads.size
print(f"r={r} steps={steps} jumps={jc.size} f_jump={jc.size / steps:.3f} slots/run={spr}")
order = np.argsort(heads, kind="stable")
hs = heads[order]
seg_lo = np.searchsorted(hs, np.arange(257))
seg_hi = seg_lo[1:]
seg_lo = seg_lo[:-1]
rc = (seg_hi - seg_lo).astype(np.float64)
lets = np.nonzero(rc > 0)[0]
# queries per letter
qcount = np.bincount(jc, minlength=256).astype(np.float64)


def uniform_geometry(max_slots, alpha=0.7):
    """spx_flatten.hip: B_c = K * share^-alpha, smallest K that fits"""
    def slots(K):
        B = np.maximum(K * (rc[lets] / r) ** -alpha, 1.0)
        return np.sum(np.floor(r / B) + 2), B
    lo, hi = 1e-6, 4e9
    for _ in range(80):
        mid = np.sqrt(lo * hi)
        if slots(mid)[0] <= max_slots:
            hi = mid
        else:
            lo = mid
    return dict(zip(lets, slots(hi)[1]))


def price_uniform(B):
    """per jump: 1 slot gather; fails when the block's first c-run lies before k: +1 when that run is the only one of the block
    (FAT_SINGLE: the next slot), +3 otherwise (fat_j, Q, dirrows)"""
    tot = {"ok": 0, "single": 0, "multi": 0}
    per_letter = {}
    for c in lets:
        sel = jc == c
        if not sel.any():
            continue
        k = jk[sel]
        Q = order[seg_lo[c]:seg_hi[c]]  # runs of the letter, ascending
        b = np.floor(k / B[c]).astype(np.int64)
        qb = np.floor(Q / B[c]).astype(np.int64)
        # first c-run of block b: first Q with qb >= b
        j0 = np.searchsorted(qb, b, side="left")
        first = np.where(j0 < Q.size, Q[np.minimum(j0, Q.size - 1)], r + 1)
        ok = first > k  # (nosucc counts as ok)
        # the block's population
        j1 = np.searchsorted(qb, b, side="right")
        pop = j1 - j0
        # the successor proper
        js = np.searchsorted(Q, k, side="left")
        single = ~ok & (pop == 1)
        multi = ~ok & (pop > 1)
        # among multi: successor still in the block?
        tot["ok"] += int(ok.sum())
        tot["single"] += int(single.sum())
        tot["multi"] += int(multi.sum())
        per_letter[c] = (k.size, int((~ok).sum()), int(multi.sum()))
    nj = jc.size
    g = (nj + tot["single"] + 3 * tot["multi"]) / nj
    return tot, g, per_letter


B = uniform_geometry(spr * r)
tot, g, per_letter = price_uniform(B)
print(f"uniform alpha=0.7: ok {tot['ok'] / jc.size:.4f} single {tot['single'] / jc.size:.4f} multi {tot['multi'] / jc.size:.4f}"
      f" -> {g:.3f} gathers per jump")
for alpha in (0.5, 0.6, 0.8, 1.0):
    t2, g2, _ = price_uniform(uniform_geometry(spr * r, alpha))
    print(f"uniform alpha={alpha}: ok {t2['ok'] / jc.size:.4f} single {t2['single'] / jc.size:.4f} multi {t2['multi'] / jc.size:.4f} -> {g2:.3f}")


# ---- piecewise geometries: every letter's run-index space cut into regions, a block size per region ----
def price_piecewise(bounds_of, alpha, max_slots, label, use_queries=False):
    """bounds_of(c) -> ascending region starts (first = 0) for letter c.  Block size of a region: K * density^-alpha (density =
    the letter's runs in the region / its length), at least 1, at most the region's length; K: smallest that fits."""
    regs = {}
    for c in lets:
        bd = np.asarray(bounds_of(c), dtype=np.int64)
        ends_ = np.append(bd[1:], r)
        L = (ends_ - bd).astype(np.float64)
        Q = order[seg_lo[c]:seg_hi[c]]
        cnt = np.diff(np.searchsorted(Q, np.append(bd, r))).astype(np.float64)
        if use_queries:
            kq = jk[jc == c]
            qn = np.diff(np.searchsorted(np.sort(kq), np.append(bd, r))).astype(np.float64)
            wgt = (cnt * (qn + 1.0)) / np.maximum(L, 1) ** 2  # q_i rho_i / L_i
        else:
            wgt = cnt / np.maximum(L, 1)
        regs[c] = (bd, L, cnt, wgt)

    def geometry(K):
        tot = 0
        out = {}
        for c in lets:
            bd, L, cnt, wgt = regs[c]
            with np.errstate(divide="ignore"):
                Bv = np.where(cnt > 0, K * np.maximum(wgt, 1e-30) ** -alpha, L)
            Bv = np.clip(Bv, 1.0, np.maximum(L, 1.0))
            nb = np.floor((L - 1) / Bv).astype(np.int64) + 1  # blocks of the region
            tot += int(nb.sum()) + 2
            out[c] = (Bv, nb)
        return tot, out

    lo, hi = 1e-9, 4e9
    for _ in range(90):
        mid = np.sqrt(lo * hi)
        if geometry(mid)[0] <= max_slots:
            hi = mid
        else:
            lo = mid
    nslots, geo = geometry(hi)
    nreg = sum(regs[c][0].size for c in lets)
    tot = {"ok": 0, "single": 0, "multi": 0}
    for c in lets:
        sel = jc == c
        if not sel.any():
            continue
        k = jk[sel]
        bd = regs[c][0]
        Bv, nb = geo[c]
        base = np.concatenate(([0], np.cumsum(nb)[:-1]))
        Q = order[seg_lo[c]:seg_hi[c]]

        def blk(x):
            ri = np.searchsorted(bd, x, side="right") - 1
            return base[ri] + np.floor((x - bd[ri]) / Bv[ri]).astype(np.int64)

        b = blk(k)
        qb = blk(Q)
        j0 = np.searchsorted(qb, b, side="left")
        first = np.where(j0 < Q.size, Q[np.minimum(j0, Q.size - 1)], r + 1)
        ok = first > k
        pop = np.searchsorted(qb, b, side="right") - j0
        tot["ok"] += int(ok.sum())
        tot["single"] += int((~ok & (pop == 1)).sum())
        tot["multi"] += int((~ok & (pop > 1)).sum())
    nj = jc.size
    g = (nj + tot["single"] + 3 * tot["multi"]) / nj
    print(f"{label}: regions {nreg} slots/run {nslots / r:.2f} ok {tot['ok'] / nj:.4f} single {tot['single'] / nj:.4f} "
          f"multi {tot['multi'] / nj:.4f} -> {g:.3f} gathers per jump", flush=True)
    return g


# F-ranges in run-index space: letter c' heads the suffixes of rows [F[c'], F[c'+1]); the run that holds F[c']
lens_ = None
if os.path.exists(cache):
    pass
frun = None


def letter_frun():
    # needs the run lengths: rebuild them cheaply from the cache when missing
    z = np.load(cache)
    if "lens" not in z:
        return None
    lens = z["lens"]
    ends = np.cumsum(lens)
    starts = ends - lens
    cnt_c = np.bincount(heads, weights=lens, minlength=256).astype(np.int64)
    F = np.concatenate(([0], np.cumsum(cnt_c)[:-1]))
    fr = np.searchsorted(starts, F, side="right") - 1
    return np.unique(np.clip(fr, 0, r - 1))


if "lens" not in np.load(cache):
    heads2, lens2, thr2, ds2, do2 = build()
    z = dict(np.load(cache))
    z["lens"] = lens2
    np.savez(cache, **z)
fr = letter_frun()
fr[0] = 0
print("letters' F-ranges (regions):", fr.size)
for alpha in (0.5, 0.7, 1.0):
    price_piecewise(lambda c: fr, alpha, spr * r, f"bigram regions alpha={alpha}")
price_piecewise(lambda c: fr, 0.5, spr * r, "bigram regions, query-weighted alpha=0.5", use_queries=True)
for G in (256, 4096, 65536):
    grid = (np.arange(G) * (r / G)).astype(np.int64)
    for alpha in (0.5, 0.7):
        price_piecewise(lambda c: grid, alpha, spr * r, f"uniform grid of {G} regions alpha={alpha}")


# ---- what the walk does with a slot: variants ----
def price_variants(B, label):
    """V0: today (one 16-byte slot; +1 next slot when FAT_SINGLE; +3 otherwise).  V1: slots b and b + 1 in one gather (same 128-byte
    line: 7 of 8): single failures cost nothing.  V2: blocks of twice the size with [first, second c-run at or after the block's start]."""
    nj = jc.size
    v0 = v1 = v2 = 0.0
    f2 = 0
    for c in lets:
        sel = jc == c
        if not sel.any():
            continue
        k = jk[sel]
        Q = order[seg_lo[c]:seg_hi[c]]
        b = np.floor(k / B[c]).astype(np.int64)
        qb = np.floor(Q / B[c]).astype(np.int64)
        j0 = np.searchsorted(qb, b, side="left")
        first = np.where(j0 < Q.size, Q[np.minimum(j0, Q.size - 1)], r + 1)
        ok = first >= k  # (k itself heads another letter: >= and > agree)
        pop = np.searchsorted(qb, b, side="right") - j0
        single = ~ok & (pop == 1)
        multi = ~ok & (pop > 1)
        v0 += k.size + single.sum() + 3 * multi.sum()
        v1 += k.size + single.sum() / 8.0 + 3 * multi.sum()
        # V2
        b2 = np.floor(k / (2 * B[c])).astype(np.int64)
        qb2 = np.floor(Q / (2 * B[c])).astype(np.int64)
        j2 = np.searchsorted(qb2, b2, side="left")
        js = np.searchsorted(Q, k, side="left")  # the successor's directory position
        fail2 = js - j2 >= 2
        f2 += int(fail2.sum())
        v2 += k.size + 3 * fail2.sum()
    print(f"{label}: V0 {v0 / nj:.3f}  V1 (two slots a gather) {v1 / nj:.3f}  V2 ([first, second] of a double block) {v2 / nj:.3f} "
          f"(fails {f2 / nj:.4f})", flush=True)


for s in (3.4, 6.8, 16.0):
    price_variants(uniform_geometry(s * r), f"uniform alpha=0.7 at {s} slots per run")


def quantile_bounds(R):
    def f(c):
        Q = order[seg_lo[c]:seg_hi[c]]
        if Q.size < 2 * R:
            return np.array([0])
        bd = Q[(np.arange(1, R) * Q.size) // R]
        return np.unique(np.concatenate(([0], bd)))
    return f


for R in (4, 16, 64):
    for alpha in (0.5, 1.0):
        price_piecewise(quantile_bounds(R), alpha, spr * r, f"quantile regions R={R} alpha={alpha}")
