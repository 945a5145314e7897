"""PCIe-inclusive rate of the host-buffer entry point spx_query_batch (never bench.py's `value`)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spumoni_amd import capi, synth

raw = synth.statistical_rlbwt(1 << 26, 253, 8.0, seed=3, device="cuda", zipf=1.0)
seqs, offs = synth.simulate_reads(raw, 10_000_000, 44, seed=13)
ix = capi.Index.from_raw(raw, 0)
hs, ho = seqs.cpu().numpy(), offs.cpu().numpy()
for rep in range(3):
    t0 = time.time()
    out = ix.query_host(capi.SPX_MODE_PML, hs, ho, classify=(150, 5))
    dt = time.time() - t0
    print(f"spx_query_batch (pageable host buffers, 0.44 GB in / 1.92 GB out): {dt*1e3:.1f} ms = {1e7/dt/1e6:.1f} M reads/s")
