"""Where tests/test_gpu_parity.py::test_long_runs_and_far_thresholds spends its time (231 s of the 436 s suite):
the same index shapes and calls, a wall-clock stamp after each (run on the GPU box)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spumoni_amd import capi, synth

T0 = time.time()
def stamp(what):
    torch.cuda.synchronize()
    print(f"{time.time() - T0:8.2f} s  {what}", flush=True)

rng = np.random.default_rng(5)
acg = np.frombuffer(b"ACG", dtype=np.uint8)
r = 4000
idx = rng.integers(0, 3, size=r)
for i in range(1, r):
    if idx[i] == idx[i - 1]:
        idx[i] = (idx[i] + 1) % 3
heads = acg[idx].copy()
lens = rng.integers(1, 6, size=r).astype(np.int64)
big = rng.random(r) < 0.05
lens[big] = rng.integers(1 << 16, 3 << 16, size=int(big.sum()))
heads[r // 2], lens[r // 2] = 0, 1
stamp("start")

def probe(raw, seqs, offs, tag):
    ix = capi.Index.from_raw(raw, 0)
    stamp(f"{tag}: Index.from_raw  {ix.describe()}")
    for what, kw in (("PML + classify", dict(classify=(7, 3))), ("PML classify only", dict(want_lengths=False, classify=(7, 3))),
                     ("PML 16", dict(classify=(7, 3), bits=16)), ("PML + doc", dict(want_docs=True)),
                     ("PML + doc 16", dict(want_docs=True, bits=16))):
        ix.query_host(capi.SPX_MODE_PML, seqs, offs, **kw)
        stamp(f"{tag}: {what}   stats {ix.last_stats()}")
    for what, kw in (("MS + doc", dict(want_lengths=False, want_docs=True)), ("MS + doc 16", dict(want_lengths=False, want_docs=True, bits=16))):
        ix.query_host(capi.SPX_MODE_MS, seqs, offs, **kw)
        stamp(f"{tag}: {what}   stats {ix.last_stats()}")
    ix.close()

raw = synth.raw_from_runs(torch.from_numpy(heads), torch.from_numpy(lens), 3, with_samples=True, n_docs=4)
stamp("(1) raw_from_runs")
seqs, offs = synth.simulate_reads(raw, 2000, 60, seed=9, positive_fraction=0.5)
stamp("(1) simulate_reads")
probe(raw, seqs.cpu().numpy(), offs.cpu().numpy(), "(1) long runs")
if time.time() - T0 < 150:
    r = (1 << 21) + 1000
    if os.environ.get("PROBE_OLD_HEADS"):  # what the test did until round 3: repair equal neighbours in place (220 s)
        idx = rng.integers(0, 3, size=r)
        eq = np.flatnonzero(idx[1:] == idx[:-1]) + 1
        while eq.size:
            idx[eq] = (idx[eq] + 1) % 3
            eq = np.flatnonzero(idx[1:] == idx[:-1]) + 1
    else:
        idx = np.cumsum(rng.integers(1, 3, size=r)) % 3
    stamp("(2) heads drawn")
    heads = acg[idx].copy()
    lens = rng.integers(1, 4, size=r).astype(np.int64)
    heads[0], lens[0] = 0, 1
    heads[500] = heads[500 + (3 << 19)] = ord("T")
    raw = synth.raw_from_runs(torch.from_numpy(heads), torch.from_numpy(lens), 1, with_samples=True, n_docs=3)
    stamp("(2) raw_from_runs")
    seqs, offs = synth.simulate_reads(raw, 20000, 40, seed=1, positive_fraction=0.5)
    stamp("(2) simulate_reads")
    seqs = seqs.cpu().numpy().copy()
    seqs[rng.random(seqs.size) < 0.05] = ord("T")
    probe(raw, seqs, offs.cpu().numpy(), "(2) far threshold")
