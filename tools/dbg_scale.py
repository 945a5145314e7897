"""Debug aid: compare the HIP walk with the oracle on a statistical index of r runs for forced fat block sizes.

    python tools/dbg_scale.py <runs> auto 0 1 2     (through gpurun)
"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from spumoni_amd import capi, synth
import oracle
r = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 25
raw = synth.statistical_rlbwt(r, 253, 8.0, seed=3, device="cuda", zipf=1.0)
seqs, offs = synth.simulate_reads(raw, 200_000, 44, seed=13)
orc = oracle.OracleIndex.from_raw(raw.cpu())
hs, ho = seqs.cpu().numpy(), offs.cpu().numpy()
want = orc.pml(hs, ho)
for bs in sys.argv[2:]:
    if bs != "auto": os.environ["SPX_FAT_BSHIFT"] = bs
    else: os.environ.pop("SPX_FAT_BSHIFT", None)
    ix = capi.Index.from_raw(raw, 0)
    got = ix.query_host(capi.SPX_MODE_PML, hs, ho)["lengths"]
    bad = np.flatnonzero(got != want)
    print("bshift", bs, "idx GiB", ix.device_bytes / 2**30, "mismatches", bad.size, "first", bad[:5], flush=True)
    if bad.size:
        q = int(bad[-1]) // 44
        print(" read", q, "chars", hs[q*44:(q+1)*44].tolist()); print(" got ", got[q*44:(q+1)*44].tolist()); print(" want", want[q*44:(q+1)*44].tolist())
    ix.close(); del ix
