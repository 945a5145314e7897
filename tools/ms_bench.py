"""MS mode end to end on a real BWT (5-strain E. coli pangenome): walk kernel vs length extension."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spumoni_amd import capi, synth

base = synth.random_genome(4_641_652, seed=1)
genomes = [base] + [synth.mutate(base, seed=s) for s in (2, 3, 4, 5)]
text, doc_lengths = synth.pangenome_text(genomes)
t0 = time.time()
raw = synth.index_from_text(torch.from_numpy(text).cuda(), doc_lengths=doc_lengths)
torch.cuda.synchronize(); print(f"index build {time.time()-t0:.1f}s  n={raw.n} r={raw.r} n/r={raw.n/raw.r:.1f}")
nreads, m = 1_000_000, 250
seqs, offs = synth.sample_reads(text, nreads, m, seed=12)
ix = capi.Index.from_raw(raw, 0)
t0 = time.time(); ix.rebuild_text(); dt = time.time() - t0
print(f"text rebuilt from the MS index (LF chains from the SA samples): {dt*1e3:.1f} ms for n = {raw.n}, identical to the text: "
      f"{bool(np.array_equal(ix.text(), text))}")
d_seqs = capi.pad_seqs(torch.from_numpy(seqs).cuda()); d_offs = torch.from_numpy(offs).cuda()
tot = nreads * m
vt = torch.int16 if os.environ.get("MS_BENCH_BITS") == "16" else torch.int32  # 16: the entry points the CLI uses
print("output width:", vt)
d_len = torch.empty(tot, dtype=vt, device="cuda"); d_ptr = torch.empty(tot, dtype=torch.int64, device="cuda")
d_doc = torch.empty(tot, dtype=vt, device="cuda"); d_cls = torch.empty((nreads, 2), dtype=torch.int64, device="cuda")
for mode, name in ((capi.SPX_MODE_PML, "PML"), (capi.SPX_MODE_MS, "MS ")):
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if mode == capi.SPX_MODE_PML:
            ix.query_device(mode, d_seqs, d_offs, tot, d_lengths=d_len, d_docs=d_doc, d_class=d_cls, bin_width=150, max_value_thr=7)
        else:
            ix.query_device(mode, d_seqs, d_offs, tot, d_lengths=d_len, d_pointers=d_ptr, d_docs=d_doc, d_class=d_cls, bin_width=150, max_value_thr=7)
        e1.record(); torch.cuda.synchronize()
        st = ix.last_stats()
    print(f"{name}+doc: total {e0.elapsed_time(e1):8.2f} ms  walk kernel {st['kernel_ms']:8.2f} ms  -> {nreads/e0.elapsed_time(e1)/1e3:7.1f} Mreads/s "
          f"f_mis {st['jumps']/st['steps']:.3f} rows/step {st['row_loads']/st['steps']:.2f} dir/step {st['dir_loads']/st['steps']:.2f}")
