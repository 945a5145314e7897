"""Burst against sustained: the headline batch walked back to back for SECONDS seconds, its time per batch printed second by
second beside the device's clocks, power and temperature (rocm-smi).  Is the walk's speed a function of how long the device has
been busy?   python tools/sustained.py [seconds=90] [runs=1000000000]"""
import os, subprocess, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spumoni_amd import capi, synth

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000_000
raw = synth.statistical_rlbwt(runs, 253, 8.0, seed=6, device="cuda", zipf=1.0)
seqs, offs = synth.simulate_reads(raw, 10_000_000, 44, seed=13, positive_fraction=0.5, f_mis=0.02, warmup=4)
total = int(seqs.numel())
ix = capi.Index.from_raw(raw, 0)
del raw
torch.cuda.empty_cache()
d_seqs = capi.pad_seqs(seqs)
d_len = torch.empty(total + 8, dtype=torch.int16, device="cuda")
d_cls = torch.empty((10_000_000, 2), dtype=torch.int64, device="cuda")


def smi():
    try:
        out = subprocess.run(["rocm-smi", "-c", "-P", "-t", "--csv"], capture_output=True, text=True, timeout=20).stdout.strip().splitlines()
        head, row = out[0].split(","), out[1].split(",")
        keep = [i for i, h in enumerate(head) if any(w in h.lower() for w in ("sclk", "mclk", "fclk", "power", "junction", "hbm", "memory)"))]
        return "  ".join(f"{head[i].strip()}={row[i].strip()}" for i in keep)
    except Exception as e:
        return "rocm-smi: " + str(e)[:80]


print("idle:", smi(), flush=True)
time.sleep(5)
t_start = time.time()
next_smi = 0.0
while time.time() - t_start < seconds:
    ms = []
    t1 = time.time()
    while time.time() - t1 < 1.0:
        ix.query_device(capi.SPX_MODE_PML, d_seqs, offs, total, d_lengths=d_len, d_class=d_cls, bin_width=150, max_value_thr=5)
        torch.cuda.synchronize()
        ms.append(ix.last_stats()["kernel_ms"])
    el = time.time() - t_start
    line = f"t={el:6.1f}s  kernel {np.median(ms):7.3f} ms (min {min(ms):.3f}, {len(ms)} batches)"
    if el >= next_smi:
        line += "  | " + smi()
        next_smi = el + 10
    print(line, flush=True)
time.sleep(20)
ix.query_device(capi.SPX_MODE_PML, d_seqs, offs, total, d_lengths=d_len, d_class=d_cls, bin_width=150, max_value_thr=5)
torch.cuda.synchronize()
for _ in range(3):
    ix.query_device(capi.SPX_MODE_PML, d_seqs, offs, total, d_lengths=d_len, d_class=d_cls, bin_width=150, max_value_thr=5)
    torch.cuda.synchronize()
    print("after 20 s of rest: kernel %.3f ms" % ix.last_stats()["kernel_ms"], flush=True)
