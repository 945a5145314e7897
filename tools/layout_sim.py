"""Gathers per searched character under candidate row layouts (CPU model; numpy, small index).

Replays the reference's search (compute_ms_pml.cpp:238-286) over a statistical RLBWT read by read and
counts, per step, what each layout would have to fetch.  The statistics that matter (how often a match
step leaves run LFrun, how often a match is followed by a jump, ...) do not depend on r, so a 2^20-run
index predicts the 10^9-run one -- except for the fat table's hit rate, which is taken as 1 gather per jump
here (the GPU counters say 1.04-1.10).

    python tools/layout_sim.py [c3|dna] [positive_fraction]
"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spumoni_amd import synth

which = sys.argv[1] if len(sys.argv) > 1 else "c3"
pf = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
R = 1 << 20
if which == "c3":
    raw = synth.statistical_rlbwt(R, 253, 8.0, seed=3, zipf=1.0)
    m, nreads, sigma = 44, 40000, 253
else:
    raw = synth.statistical_rlbwt(R, 4, 60.0, seed=4, letters=b"ACGT")
    m, nreads, sigma = 200, 10000, 4
w = 1
while sigma ** w < R:
    w += 1
seqs, offs = synth.simulate_reads(raw, nreads, m, seed=13, positive_fraction=pf, f_mis=0.02, warmup=w)
seqs = seqs.numpy().reshape(nreads, m).astype(np.int64)

heads = raw.heads.numpy().astype(np.int64); heads[heads <= 1] = 1
lens = raw.lens.numpy(); r = heads.size
ends = np.cumsum(lens); starts = ends - lens; n = int(ends[-1])
order = np.argsort(heads, kind="stable")
lf_sorted = np.cumsum(lens[order]) - lens[order]
lf_start = np.empty(r, np.int64); lf_start[order] = lf_sorted        # LF(start of run k)
LFrun = np.searchsorted(starts, lf_start, side="right") - 1
LFoff = lf_start - starts[LFrun]
hs = heads[order]; comp = hs * r + order
seg_lo = np.searchsorted(hs, np.arange(257)); seg_hi = seg_lo[1:]; seg_lo = seg_lo[:-1]
thr = raw.thr.numpy()

pos = np.full(nreads, n - 1, np.int64)
# per step records
rec = []
for i in range(m):
    c = seqs[:, m - 1 - i]
    k = np.minimum(np.searchsorted(starts, pos, side="right") - 1, r - 1)
    off = pos - starts[k]
    at_end = pos >= n
    present = seg_hi[c] > seg_lo[c]
    match = (heads[k] == c) & ~at_end & (c < 128) & present
    quirk_stay = (heads[k] == c) & ~at_end & (c >= 128) & present   # byte >= 128 on its own run: jump branch, stays
    kk = np.where(at_end, r, k)
    jg = np.searchsorted(comp, c * r + kk)
    has_succ = jg < seg_hi[c]
    qs = order[np.minimum(jg, r - 1)]
    th = np.where(has_succ, thr[qs], n + 1)
    th = np.where(has_succ & (jg == seg_lo[c]), 0, th)
    use_pred = pos < th
    qp = order[np.maximum(jg - 1, 0)]
    q = np.where(use_pred, qp, qs)
    jpos = np.where(use_pred, starts[qp] + lens[qp] - 1, starts[qs])
    stay = match | quirk_stay
    run = np.where(stay, k, q)
    newpos = np.where(stay, pos, jpos)
    # absent letters: LF = F[c]: ignore (none in these reads)
    npos = lf_start[run] + (newpos - starts[run])
    k0 = np.minimum(np.searchsorted(starts, npos, side="right") - 1, r - 1)
    t = k0 - LFrun[run]
    rec.append((match.copy(), t.copy(), k0.copy(), run.copy()))
    pos = npos

steps = nreads * m
M = np.array([x[0] for x in rec])      # [m, nreads] match flags
T = np.array([x[1] for x in rec])
nxt_match = np.zeros_like(M); nxt_match[:-1] = M[1:]
last = np.zeros_like(M); last[-1] = True
f_mis = 1 - M.mean()
print(f"{which} positive_fraction={pf}: steps {steps}, f_mis {f_mis:.3f}; match steps landing in LFrun+t: "
      + ", ".join(f"t={j}: {((T == j) & M).sum() / max(1, M.sum()):.3f}" for j in range(5)))
# heads of landing runs (for the peek): jump lands in run k0 with known head (current layout) -> row gather only if next is a match
J = ~M
# --- S0, the shipped layout: match -> row gather; jump -> 1 fat gather, + row gather iff the next step is a match
row0 = (M & ~last).sum() + (J & nxt_match).sum()
fat = J.sum()
print(f"S0 shipped      : row {row0 / steps:.3f} + fat {fat / steps:.3f} = {(row0 + fat) / steps:.3f} gathers/char, "
      f"lane loads {(row0 + fat) / steps:.3f}")
# --- S1: row also carries the heads of its 4 destination runs: match followed by a jump needs no row gather (t < 4)
row1 = (M & nxt_match & ~last).sum() + (M & ~nxt_match & ~last & (T >= 4)).sum() + (J & nxt_match).sum()
print(f"S1 +dest heads  : row {row1 / steps:.3f} + fat {fat / steps:.3f} = {(row1 + fat) / steps:.3f} gathers/char")
# --- S2: 32-byte rows that embed the lite row of ONE destination (the most likely t, here t = 0) + heads of the others
# state: after a row gather we hold (full row incl. embedded); after stepping onto the embedded row we hold a lite row
# (heads of ITS destinations known, no embedded row)
def sim_s2(embed_t_of_run, heads_known=True):
    full = np.zeros(nreads, bool)   # standing on a row fetched from memory (embedded destination available)
    full[:] = True                  # the initial row comes with the kernel arguments
    rowg = 0
    for i in range(m):
        Mi, Ti, k0i, runi = rec[i]
        nm = rec[i + 1][0] if i + 1 < m else np.zeros(nreads, bool)
        lastc = i + 1 == m
        if lastc:
            break
        # match step from a full row whose embedded destination is the one we land in: no gather, now on a lite row
        emb = Mi & full & (Ti == embed_t_of_run[runi])
        # otherwise after a match: need row k0 iff next is a match, or (next is a jump and head unknown)
        need = Mi & ~emb & (nm | (~nm & ((Ti >= 4) | (not heads_known))))
        # after a jump: row gather iff the next step is a match
        needj = ~Mi & nm
        g = need | needj
        rowg += g.sum()
        # new state: full if gathered; lite if stepped onto embedded; after a jump w/o gather: no row at all (next is a jump)
        full = g
    return rowg
emb0 = np.zeros(r, np.int64)
row2 = sim_s2(emb0)
print(f"S2 32B rows, embed t=0 : row {row2 / steps:.3f} (two lane loads each) + fat {fat / steps:.3f} = "
      f"{(row2 + fat) / steps:.3f} gathers/char, lane loads {(2 * row2 + fat) / steps:.3f}")
# best t per run: the destination holding most of the run's offsets
cum = np.zeros((r, 5), np.int64)
room = lens[LFrun] - LFoff
cum[:, 0] = room
for j in range(1, 5):
    cum[:, j] = cum[:, j - 1] + lens[np.minimum(LFrun + j, r - 1)]
cov = np.minimum(cum, lens[:, None])
share = np.diff(np.concatenate([np.zeros((r, 1), np.int64), cov], 1), axis=1)[:, :4]
embb = share.argmax(1)
row2b = sim_s2(embb)
print(f"S2 32B rows, embed best: row {row2b / steps:.3f} + fat {fat / steps:.3f} = {(row2b + fat) / steps:.3f} gathers/char, "
      f"lane loads {(2 * row2b + fat) / steps:.3f}")
row2n = sim_s2(emb0, heads_known=False)
print(f"S2 32B rows, embed t=0, NO destination heads: row {row2n / steps:.3f} + fat {fat / steps:.3f} = {(row2n + fat) / steps:.3f} gathers/char, "
      f"lane loads {(2 * row2n + fat) / steps:.3f}")


# --- S3: wider rows that embed the lite rows of TWO destinations (t = 0 and t = 1), heads of the others
def sim_s3(ts=(0, 1)):
    full = np.ones(nreads, bool)
    rowg = 0
    for i in range(m - 1):
        Mi, Ti, k0i, runi = rec[i]
        nm = rec[i + 1][0]
        emb = Mi & full & np.isin(Ti, ts)
        need = Mi & ~emb & (nm | (~nm & (Ti >= 4)))
        needj = ~Mi & nm
        g = need | needj
        rowg += g.sum()
        full = g
    return rowg


# --- S4: 64-byte rows that embed destination t = 0 WITH that row's own embedded destination (a chain of two):
# up to three LF steps per gather
def sim_s4():
    depth = np.full(nreads, 2)  # embedded levels still available below the row the lane stands on
    rowg = 0
    for i in range(m - 1):
        Mi, Ti, k0i, runi = rec[i]
        nm = rec[i + 1][0]
        emb = Mi & (depth > 0) & (Ti == 0)
        need = Mi & ~emb & (nm | (~nm & (Ti >= 4)))
        needj = ~Mi & nm
        g = need | needj
        rowg += g.sum()
        depth = np.where(g, 2, np.where(emb, depth - 1, 0))
    return rowg


row3 = sim_s3()
print(f"S3 48B rows, embed t=0 and t=1: row {row3 / steps:.3f} + fat {fat / steps:.3f} = {(row3 + fat) / steps:.3f} gathers/char, "
      f"lane loads {(3 * row3 + fat) / steps:.3f}")
row3b = sim_s3((0, 1, 2))
print(f"S3' 64B rows, embed t=0,1,2  : row {row3b / steps:.3f} + fat {fat / steps:.3f} = {(row3b + fat) / steps:.3f} gathers/char, "
      f"lane loads {(4 * row3b + fat) / steps:.3f}")
row4 = sim_s4()
print(f"S4 48B rows, chain of two t=0 : row {row4 / steps:.3f} + fat {fat / steps:.3f} = {(row4 + fat) / steps:.3f} gathers/char, "
      f"lane loads {(3 * row4 + fat) / steps:.3f}")
