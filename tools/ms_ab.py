"""The MS walk on the C4 shape, per output set and occupancy (run through gpurun; SPUMONI_GPU_LIB picks the build).
   python tools/ms_ab.py [log2 runs = 27]
Index: statistical RLBWT r = 2^k, sigma 253, SA samples + 10 documents; 5*10^6 x 55 characters, 16-bit values.
Prints kernel ms (library events) for PML / PML+doc / MS / MS+doc at 12 / 16 / 20 waves per CU, and checks the first
5000 reads of MS+doc against the CPU restatement."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spumoni_amd import capi, synth

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 27
raw = synth.statistical_rlbwt(1 << lg, 253, 8.0, seed=5, device="cuda", zipf=1.0, with_samples=True, n_docs=10)
seqs, offs = synth.simulate_reads(raw, 5_000_000, 55, seed=15, warmup=4)
gate = None
if not os.environ.get("MS_AB_NOGATE"):
    import oracle
    orc = oracle.OracleIndex.from_raw(raw.cpu())
    S = 5000
    gate = orc.ms(seqs[: S * 55].cpu().numpy(), offs[: S + 1].cpu().numpy(), want_docs=True)
ix = capi.Index.from_raw(raw, 0)
del raw
total = int(seqs.numel()); nreads = offs.numel() - 1
d_seqs = capi.pad_seqs(seqs)
d_len = torch.empty(total + 8, dtype=torch.int16, device="cuda")
d_ptr = torch.empty(total + 8, dtype=torch.int64, device="cuda")
d_doc = torch.empty(total + 8, dtype=torch.int16, device="cuda")
tag = os.path.basename(os.environ.get("SPUMONI_GPU_LIB", "libspumoni_gpu.so"))
only = os.environ.get("MS_AB_MODES")  # e.g. "MS+doc" (counter runs: one kernel variant)
waves = [int(x) for x in os.environ.get("MS_AB_WAVES", "12,16,20").split(",")]  # 0 = the library's own choice
for name, mode, kw in (("PML", capi.SPX_MODE_PML, dict(d_lengths=d_len)),
                       ("PML+doc", capi.SPX_MODE_PML, dict(d_lengths=d_len, d_docs=d_doc)),
                       ("MS", capi.SPX_MODE_MS, dict(d_pointers=d_ptr)),
                       ("MS+doc", capi.SPX_MODE_MS, dict(d_pointers=d_ptr, d_docs=d_doc))):
    line = []
    if only and name not in only.split(","):
        continue
    for w in waves:
        ix.set_option("waves_per_cu", w)
        ms = []
        for _ in range(4):
            ix.query_device(mode, d_seqs, offs, total, **kw)
            torch.cuda.synchronize()
            ms.append(ix.last_stats()["kernel_ms"])
        k = float(np.median(ms[1:]))
        line.append(f"{w}w {k:6.2f} ms {total / k / 1e6:5.1f} G/s")
    st = ix.last_stats()
    print(f"{tag:28s} {name:8s} " + " | ".join(line) + f"  f_mis {st['jumps'] / st['steps']:.3f} rows {st['row_loads'] / st['steps']:.2f} dir {st['dir_loads'] / st['steps']:.2f}", flush=True)
if gate is not None:
    S = 5000
    ok = (np.array_equal(d_ptr[: S * 55].cpu().numpy().view(np.uint64), gate["pointers"]) and
          np.array_equal(d_doc[: S * 55].cpu().numpy().view(np.uint16).astype(np.uint32), gate["docs"]))
    print(f"{tag:28s} MS+doc == CPU restatement on {S} reads: {ok}", flush=True)
