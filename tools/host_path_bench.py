"""PCIe-inclusive rate of the host-buffer entry point spx_query_batch (never bench.py's `value`)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spumoni_amd import capi, synth

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 26
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # random warm-up characters of the positive reads (bench.py: 4)
raw = synth.statistical_rlbwt(runs, 253, 8.0, seed=3, device="cuda", zipf=1.0)
seqs, offs = synth.simulate_reads(raw, 10_000_000, 44, seed=13, warmup=warm)
print(f"index r = {runs}, reads with {warm} warm-up characters")
ix = capi.Index.from_raw(raw, 0)
del raw; torch.cuda.empty_cache()
hs, ho = seqs.cpu().numpy(), offs.cpu().numpy()
for rep in range(3):
    t0 = time.time()
    out = ix.query_host(capi.SPX_MODE_PML, hs, ho, classify=(150, 5))
    dt = time.time() - t0
    print(f"spx_query_batch (pageable host buffers, 0.44 GB in / 1.92 GB out): {dt*1e3:.1f} ms = {1e7/dt/1e6:.1f} M reads/s")

# the same with page-locked buffers from spx_host_alloc
import ctypes as C
tot = hs.size; nreads = ho.size - 1
ps, o1 = capi.pinned_array((tot,), np.uint8); ps[:] = hs
po, o2 = capi.pinned_array((nreads + 1,), np.uint64); po[:] = ho
pl, o3 = capi.pinned_array((tot,), np.uint32)
pc, o4 = capi.pinned_array((nreads,), capi.CLASS_DTYPE)
vp = lambda a: a.ctypes.data_as(C.c_void_p)
for rep in range(3):
    t0 = time.time()
    rc = capi.lib().spx_query_batch(ix._h, capi.SPX_MODE_PML, vp(ps), vp(po), nreads, vp(pl), None, None, vp(pc), 150, 5)
    dt = time.time() - t0
    assert rc == 0
    print(f"spx_query_batch (page-locked buffers): {dt*1e3:.1f} ms = {1e7/dt/1e6:.1f} M reads/s")
assert np.array_equal(pl, out["lengths"])

# 16-bit outputs (spx_query_batch16): half the bytes to copy back
pl16, o5 = capi.pinned_array((tot + 8,), np.uint16)
for rep in range(3):
    t0 = time.time()
    rc = capi.lib().spx_query_batch16(ix._h, capi.SPX_MODE_PML, vp(ps), vp(po), nreads, vp(pl16), None, None, vp(pc), 150, 5)
    dt = time.time() - t0
    assert rc == 0
    print(f"spx_query_batch16 (page-locked buffers, 0.44 GB in / 1.04 GB out): {dt*1e3:.1f} ms = {1e7/dt/1e6:.1f} M reads/s")
assert np.array_equal(pl16[:tot], out["lengths"])
