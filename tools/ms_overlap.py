"""Experiment: does the MS length extension overlap with the MS walk when the batch is cut in parts that run on two
streams (two index handles: a handle orders its own queries)?  E. coli case of tools/ms_bench.py, 16-bit outputs."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spumoni_amd import capi, synth

base = synth.random_genome(4_641_652, seed=1)
genomes = [base] + [synth.mutate(base, seed=s) for s in (2, 3, 4, 5)]
text, doc_lengths = synth.pangenome_text(genomes)
raw = synth.index_from_text(torch.from_numpy(text).cuda(), doc_lengths=doc_lengths)
nreads, m = 1_000_000, 250
seqs, offs = synth.sample_reads(text, nreads, m, seed=12)
ix = capi.Index.from_raw(raw, 0)
ix.rebuild_text()
ix2 = ix.clone(0)
d_seqs = capi.pad_seqs(torch.from_numpy(seqs).cuda()); d_offs = torch.from_numpy(offs).cuda()
tot = nreads * m
d_len = torch.empty(tot, dtype=torch.int16, device="cuda"); d_ptr = torch.empty(tot, dtype=torch.int64, device="cuda")
d_doc = torch.empty(tot, dtype=torch.int16, device="cuda"); d_cls = torch.empty((nreads, 2), dtype=torch.int64, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(parts):
    cuts = [nreads * i // parts for i in range(parts + 1)]
    offs_parts = [d_offs[cuts[i]:cuts[i + 1] + 1].contiguous() for i in range(parts)]  # absolute offsets: results land in place
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(parts):
        lo, hi = cuts[i], cuts[i + 1]
        a, b = lo * m, hi * m
        h, st = (ix, s1) if i % 2 == 0 else (ix2, s2)
        h.query_device(capi.SPX_MODE_MS, d_seqs, offs_parts[i], b - a, d_lengths=d_len, d_pointers=d_ptr,
                       d_docs=d_doc, d_class=d_cls[lo:hi], bin_width=150, max_value_thr=7, stream=st)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0)


ref = None
for parts in (1, 2, 4, 8, 16, 1):
    ts = [run(parts) for _ in range(5)]
    key = (d_len.sum().item(), d_ptr.sum().item(), d_doc.sum().item(), d_cls.sum().item())
    ref = ref or key
    print(f"parts {parts:2d}: wall ms {' '.join(f'{t:6.2f}' for t in ts)}   same results: {key == ref}")
