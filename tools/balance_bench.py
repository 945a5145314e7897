"""Balanced pieces on a heavy-tailed index: the same reads on the same run list flattened with SPX_BALANCE_SPAN = 0
(no balancing), 64, 16 (the default) and 8.  Run on the GPU box:  python tools/balance_bench.py"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spumoni_amd import capi, synth

dev = torch.device("cuda", 0)
rng = np.random.default_rng(8)
r = int(os.environ.get("BAL_RUNS", str(1 << 21)))
nreads, m = int(os.environ.get("BAL_READS", "400000")), 100
idx = np.cumsum(rng.integers(1, 4, size=r)) % 4
lens = np.minimum((rng.pareto(1.2, size=r) + 1).astype(np.int64), 1 << 20)
heads = np.frombuffer(b"ACGT", dtype=np.uint8)[idx].copy()
heads[r // 2], lens[r // 2] = 0, 1
raw = synth.raw_from_runs(torch.from_numpy(heads).to(dev), torch.from_numpy(lens).to(dev), 4)
seqs, offs = synth.simulate_reads(raw, nreads, m, seed=2, positive_fraction=0.5, f_mis=0.02, warmup=11)
total = int(seqs.numel())
d_seqs = capi.pad_seqs(seqs)
print(f"Pareto(1.2) run lengths, sigma 4: r {r}, n {int(lens.sum())}, longest run {int(lens.max())}; {nreads} reads x {m}, half simulated-positive", flush=True)
ref = None
for combo in os.environ.get("BAL_SPANS", "0,64,16,8").split(","):  # span or span:passes
    span, _, passes = combo.partition(":")
    os.environ["SPX_BALANCE_SPAN"] = span
    if passes:
        os.environ["SPX_BALANCE_PASSES"] = passes
        span = combo
    t0 = time.time()
    ix = capi.Index.from_raw(raw, 0)
    torch.cuda.synchronize()
    t_flat = time.time() - t0
    d_len = torch.empty(total + 8, dtype=torch.int16, device=dev)
    d_cls = torch.empty((nreads, 2), dtype=torch.int64, device=dev)
    ms = []
    for rep in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ix.query_device(capi.SPX_MODE_PML, d_seqs, offs, total, d_lengths=d_len, d_class=d_cls, bin_width=150, max_value_thr=5)
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    st = ix.last_stats()
    d = ix.describe()
    same = "" if ref is None else f", same values: {bool(torch.equal(ref, d_len[:total]))}"
    if ref is None:
        ref = d_len[:total].clone()
    best = float(np.median(ms[1:]))
    print(f"span {span:>4}: rows {d['flat_runs']} (+{100.0 * (d['flat_runs'] - r) / r:.2f} %), flatten {t_flat:.2f} s, walk {best:.3f} ms = "
          f"{st['steps'] / best / 1e6:.2f} G steps/s, row gathers/step {st['row_loads'] / st['steps']:.3f}, dir/step {st['dir_loads'] / st['steps']:.3f}{same}", flush=True)
    ix.close()
