"""Round 5: what selects the regime of the walk over the r = 2e9 index (profiles/r04_c5_variance.txt: the same index walks 10^7 x 44
reads in 12.1 / 13.2 / 13.6-16.5 ms "depending on the run"; clocks under load, fragmentation at >= 2 MB, leg and gate order, the
box were cleared).  Two questions, one process:
  A  is it the device's state in TIME (idle cools the HBM stacks; refresh rate, fabric clocks)?  One index; the batch timed right
     after the flatten, through 25 s of back-to-back walking, after 75 s of idling, through 25 s more;
  B  is it WHERE the index lies?  The index laid out again and again in the same process -- at once, after an idle gap, after a
     100 GB block was allocated and freed, with the raw arrays freed before / after -- the arrays' addresses printed beside the
     batch time.
    python tools/c5_regimes.py [runs=2000000000]"""
import os, subprocess, sys, time
os.environ["SPX_DESCRIBE_ADDRESSES"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spumoni_amd import capi, synth

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000_000
NREADS = 10_000_000


def smi():
    try:
        out = subprocess.run(["rocm-smi", "-c", "-P", "-t", "--csv"], capture_output=True, text=True, timeout=20).stdout.strip().splitlines()
        head, row = out[0].split(","), out[1].split(",")
        keep = [i for i, h in enumerate(head) if any(w in h.lower() for w in ("sclk", "mclk", "fclk", "power", "junction", "hbm", "memory)"))]
        return "  ".join(f"{head[i].strip()}={row[i].strip()}" for i in keep)
    except Exception as e:
        return "rocm-smi: " + str(e)[:80]


def make_raw():
    return synth.statistical_rlbwt(runs, 253, 8.0, seed=6, device="cuda", zipf=1.0)


raw = make_raw()
seqs, offs = synth.simulate_reads(raw, NREADS, 44, seed=13, positive_fraction=0.5, f_mis=0.02, warmup=4)
total = int(seqs.numel())
d_seqs = capi.pad_seqs(seqs)
d_len = torch.empty(total + 8, dtype=torch.int16, device="cuda")
d_cls = torch.empty((NREADS, 2), dtype=torch.int64, device="cuda")


def batch(ix, reps=4):
    ms = []
    for _ in range(reps):
        ix.query_device(capi.SPX_MODE_PML, d_seqs, offs, total, d_lengths=d_len, d_class=d_cls, bin_width=150, max_value_thr=5)
        torch.cuda.synchronize()
        ms.append(ix.last_stats()["kernel_ms"])
    return ms


def where(ix):
    d = ix.describe()
    return f"rows {d.get('rows_at')} dirrows {d.get('dirrows_at')} fat {d.get('fat_at')} slots/run {d['fat_slots_per_run']:.3f} free {torch.cuda.mem_get_info()[0] / 2**30:.0f} GiB"


print("idle:", smi(), flush=True)
del seqs
torch.cuda.empty_cache()  # (the read simulation's per-run tables are as large as the raw index)
if os.environ.get("C5_REGIMES_PART_A"):
    # ---- A: one index, time and temperature ----
    t0 = time.time()
    ix = capi.Index.from_raw(raw, 0)
    torch.cuda.synchronize()
    print(f"A: flattened in {time.time() - t0:.1f} s | {where(ix)}", flush=True)
    print("A: right after the flatten:", " ".join(f"{x:.2f}" for x in batch(ix, 6)), "ms |", smi(), flush=True)
    for phase, (busy, idle) in enumerate([(25, 75), (25, 0)]):
        t1 = time.time()
        while time.time() - t1 < busy:
            ms = batch(ix, 40)
            print(f"A: busy t={time.time() - t1:5.1f}s  median {np.median(ms):.2f} (min {min(ms):.2f} max {max(ms):.2f}) |", smi(), flush=True)
        if idle:
            time.sleep(idle)
            print(f"A: after {idle} s idle:", " ".join(f"{x:.2f}" for x in batch(ix, 6)), "ms |", smi(), flush=True)
    ix.close()
# ---- B: the index laid out again, under different circumstances ----
class DevArray:  # a library-owned device array as a torch tensor (no copy)
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes // 8,), "typestr": "<i8", "data": (ptr, False), "version": 2}


def probe(ix):
    if os.environ.get("C5_REGIMES_HOLDS"):
        return ""
    """independent random 8-byte reads, one per 16-byte record, over each of the three big arrays (torch.take): does the memory
    system itself answer differently where the walk does?"""
    try:
        d = ix.describe()
        out = []
        r = int(d["flat_runs"])
        for name, nbytes in (("rows_at", r * 32), ("dirrows_at", r * 32), ("fat_at", int(d["fat_slots"]) * int(d["fat_stride"]))):
            t = torch.as_tensor(DevArray(int(d[name], 16), nbytes), device="cuda")
            idx = torch.randint(0, t.shape[0] // 2, (1 << 25,), device="cuda") * 2
            acc = torch.take(t, idx).sum()  # warm-up
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                acc = acc + torch.take(t, idx).sum()
            e1.record()
            torch.cuda.synchronize()
            out.append(f"{name[:-3]} {4 * (1 << 25) / (e0.elapsed_time(e1) * 1e-3) / 1e9:.1f}")
            del t, idx
        return "torch.take G/s: " + ", ".join(out)
    except Exception as e:
        return "probe failed: " + str(e).splitlines()[0][:100]


def lay_out(tag, before=None, free_raw_first=False, hold_gb=0):
    global raw
    if before:
        before()
    torch.cuda.empty_cache()
    held = torch.empty(hold_gb << 30, dtype=torch.uint8, device="cuda") if hold_gb else None  # (shifts where the arrays land)
    t0 = time.time()
    ix = capi.Index.from_raw(raw, 0)
    torch.cuda.synchronize()
    del held
    torch.cuda.empty_cache()
    if free_raw_first:
        raw_keep = raw
        raw = None
        del raw_keep
        torch.cuda.empty_cache()
    ms = batch(ix, 6)
    print(f"B: {tag}: {np.median(ms):.2f} ms (min {min(ms):.2f}) | flatten {time.time() - t0:.1f} s | {where(ix)} | {probe(ix)}", flush=True)
    ix.close()
    if raw is None:
        raw = make_raw()


def big_block():
    x = torch.empty(100 * 2**30, dtype=torch.uint8, device="cuda")
    x.fill_(1)
    torch.cuda.synchronize()
    del x
    torch.cuda.empty_cache()


if os.environ.get("C5_REGIMES_PART_A"):
    lay_out("again, at once")
    lay_out("after 60 s idle", before=lambda: time.sleep(60))
    lay_out("after a 100 GB block was allocated, written and freed", before=big_block)
if os.environ.get("C5_REGIMES_HOLDS"):  # (tools/c5_regimes_pmc.sh: the two regimes, nothing else)
    for gb in [int(x) for x in os.environ["C5_REGIMES_HOLDS"].split(",")]:
        lay_out(f"{gb} GB held during the layout", hold_gb=gb)
    sys.exit(0)
for gb in (0, 4, 0, 12, 1, 24, 0, 2):
    lay_out(f"{gb} GB held during the layout", hold_gb=gb)
lay_out("raw arrays freed before the batch", free_raw_first=True)
for gb in (0, 6, 0):
    lay_out(f"{gb} GB held during the layout (new raw arrays)", hold_gb=gb)
