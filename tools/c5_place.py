"""Does WHERE the index arrays were allocated matter at r = 2e9?  The same index flattened (a) from raw arrays on the device
(the bench's way: 34 GB of raw arrays + torch's cache alive while the index is laid out) and (b) from raw arrays on the host
after everything on the device was released; the headline batch on both.   python tools/c5_place.py [runs]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spumoni_amd import capi, synth

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000_000
raw = synth.statistical_rlbwt(runs, 253, 8.0, seed=6, device="cuda", zipf=1.0)
seqs, offs = synth.simulate_reads(raw, 10_000_000, 44, seed=13, positive_fraction=0.5, f_mis=0.02, warmup=4)
total = int(seqs.numel())


def bench(ix, tag, settle=0.0):
    time.sleep(settle)
    d_seqs = capi.pad_seqs(seqs)
    d_len = torch.empty(total + 8, dtype=torch.int16, device="cuda")
    d_cls = torch.empty((10_000_000, 2), dtype=torch.int64, device="cuda")
    ms = []
    for _ in range(5):
        ix.query_device(capi.SPX_MODE_PML, d_seqs, offs, total, d_lengths=d_len, d_class=d_cls, bin_width=150, max_value_thr=5)
        torch.cuda.synchronize()
        ms.append(ix.last_stats()["kernel_ms"])
    st = ix.last_stats()
    free, tot = torch.cuda.mem_get_info()
    print(f"{tag:60s} kernel {np.median(ms[1:]):7.3f} ms  dir/step {st['dir_loads'] / st['steps']:.3f}  slots/run {ix.describe()['fat_slots_per_run']}  "
          f"device free {free / 1e9:.0f} GB", flush=True)
    return d_len[:total].clone()


torch.cuda.empty_cache()
ix = capi.Index.from_raw(raw, 0)
a = bench(ix, "flattened from device arrays (raw + torch cache alive)")
a = bench(ix, "  the same index again, 3 s later", 3.0)
ix.close()
raw_h = raw.cpu()
del raw, ix
torch.cuda.empty_cache()
time.sleep(2)
ix = capi.Index.from_raw(raw_h, 0)
b = bench(ix, "flattened from host arrays (device empty before)")
b = bench(ix, "  the same index again, 3 s later", 3.0)
print("same values:", bool(torch.equal(a, b)))
