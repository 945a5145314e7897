"""The read mix from all-positive to all-random on the declared C3 index (the headline's 50 / 50 mix is
jump-heavy; reads a classifier calls FOUND are match-heavy).  One index, five batches of 10^7 x 44; per batch the
walk's rate, the gathers per character and the SURVEY 8(d) roofline fraction.  Run through gpurun."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from spumoni_amd import capi, synth

runs = int(os.environ.get("MIX_RUNS", "1000000000"))
dev = torch.device("cuda", 0)
raw = synth.statistical_rlbwt(runs, 253, 8.0, seed=3, device=dev, zipf=1.0)
w = 1
while 253 ** w < runs:
    w += 1
batches = {pf: synth.simulate_reads(raw, 10_000_000, 44, seed=31, positive_fraction=pf, f_mis=0.02, warmup=w) for pf in (1.0, 0.75, 0.5, 0.25, 0.0)}
torch.cuda.empty_cache()
ix = capi.Index.from_raw(raw, 0)
del raw
torch.cuda.empty_cache()
print(json.dumps(ix.describe()))


class A:  # what bench.walk_leg reads of the arguments
    out_bits, bin_width, max_value_thr, extra_steps = 16, 150, 5, 5


import time
time.sleep(2.0)
print(f"{'positive':>8s} {'M reads/s':>10s} {'G steps/s':>10s} {'frac':>6s} {'f_mis':>6s} {'rows/ch':>8s} {'fat+dir/ch':>10s} {'B/step':>7s}")
for pf, (s, o) in batches.items():
    r = bench.walk_leg(A, torch, capi, ix, s, o, dev, "")
    print(f"{pf:8.2f} {r['value'] / 1e6:10.1f} {r['steps_per_s'] / 1e9:10.2f} {r['roofline_frac']:6.3f} {r['f_mis']:6.3f} "
          f"{r['row_loads_per_step']:8.3f} {r['dir_loads_per_step']:10.3f} {r['bytes_per_step']:7.1f}", flush=True)
