"""The walk on a REAL digested BWT at the declared C3 table density, one library build per process (A/B: tools/real_ab.sh).

Stage 1 (once per box, cached in /dev/shm): 10 haplotypes of a REAL_AB_GENOME_BP genome (default 20 Mbp) + reverse complements,
digested -m k=4 w=11, index of the digested text built on the GPU; 10^7 x 200 bp DNA reads (half from the text with 1 %
substitutions, half reversed).  Stage 2: flatten at SPX_FAT_SLOTS_PER_RUN (default 6.8 = what the declared C3 gets at r = 10^9),
spx_digest_query_batch_device16, HIP events; oracle gate on REAL_AB_CHECK reads (default 4000; 0 = none).

Prints one line: G steps/s of the walk kernel, gathers per character.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SPX_FAT_SLOTS_PER_RUN", "6.8")
from spumoni_amd import capi, synth  # noqa: E402

genome_bp = int(float(os.environ.get("REAL_AB_GENOME_BP", "20000000")))
nreads = int(float(os.environ.get("REAL_AB_READS", "10000000")))
ncheck = int(os.environ.get("REAL_AB_CHECK", "4000"))
steps = int(os.environ.get("REAL_AB_STEPS", "5"))
bp, k, w = 200, 4, 11
cache = f"/dev/shm/real_ab_{genome_bp}_{nreads}.npz"
dev = torch.device("cuda:0")


def stage1():
    base = synth.random_genome(genome_bp, seed=1)
    genomes = [base] + [synth.mutate(base, seed=sd) for sd in range(2, 11)]
    dig = capi.digester(0)
    parts, dna = [], []
    for g in genomes:
        for seq in (g, synth.revcomp(g)):
            d, _ = dig.digest_host(capi.SPX_DIGEST_PROMOTED, k, w, seq, np.array([0, seq.size], dtype=np.uint64))
            parts.append(d.copy())
            dna.append(seq)
    dig.close()
    dtext = np.concatenate(parts)
    text = torch.from_numpy(np.concatenate(dna)).to(dev)
    raw = synth.index_from_text(torch.from_numpy(dtext).to(dev), with_samples=False)
    g = torch.Generator(device=dev)
    g.manual_seed(12)
    start = torch.randint(0, int(text.numel()) - bp, (nreads,), generator=g, device=dev)
    null = torch.rand(nreads, generator=g, device=dev) < 0.5
    reads = torch.empty((nreads, bp), dtype=torch.uint8, device=dev)
    ar = torch.arange(bp, device=dev)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    CH = 1 << 20
    for lo in range(0, nreads, CH):
        hi = min(nreads, lo + CH)
        blk = text[start[lo:hi, None] + ar[None, :]]
        sub = torch.rand((hi - lo, bp), generator=g, device=dev) < 0.01
        blk = torch.where(sub, acgt[torch.randint(0, 4, (hi - lo, bp), generator=g, device=dev)], blk)
        blk = torch.where(null[lo:hi, None], torch.flip(blk, [1]), blk)
        reads[lo:hi] = blk
    np.savez(cache, heads=raw.heads.cpu().numpy(), lens=raw.lens.cpu().numpy(), thr=raw.thr.cpu().numpy(), n=raw.n,
             reads=reads.reshape(-1).cpu().numpy())


if not os.path.exists(cache):
    t0 = time.time()
    stage1()
    print(f"[real_ab] built {cache} in {time.time() - t0:.1f}s", file=sys.stderr)
    torch.cuda.empty_cache()
z = np.load(cache)
raw = synth.RawIndex(heads=torch.from_numpy(z["heads"]), lens=torch.from_numpy(z["lens"]), thr=torch.from_numpy(z["thr"]), n=int(z["n"]))
ix = capi.Index.from_raw(raw, 0)
desc = ix.describe()
d_reads = torch.zeros(nreads * bp + 64, dtype=torch.uint8, device=dev)
d_reads[: nreads * bp] = torch.from_numpy(z["reads"]).to(dev)
d_offs = torch.arange(nreads + 1, dtype=torch.int64, device=dev) * bp
d_l16 = torch.empty(nreads * bp + 64, dtype=torch.int16, device=dev)
d_cls = torch.empty((nreads, 2), dtype=torch.int64, device=dev)
keep = {}


def both():
    d_oo, keep["work"] = ix.digest_query_device(capi.SPX_MODE_PML, capi.SPX_DIGEST_PROMOTED, k, w, d_reads, d_offs, nreads * bp,
                                                d_lengths=d_l16, d_class=d_cls, bin_width=50, max_value_thr=5, work=keep.get("work"))
    keep["oo"] = d_oo


both()
torch.cuda.synchronize()
kms = []
for _ in range(steps):
    both()
    torch.cuda.synchronize()
    kms.append(ix.last_stats()["kernel_ms"])
st = ix.last_stats()
km = float(np.median(kms))
line = (f"{os.path.basename(capi.LIB_PATH)} slots/run {desc['fat_slots_per_run']:.2f} r {raw.r}: walk {km:.3f} ms "
        f"{st['steps'] / km / 1e6:.2f} G steps/s | rows/step {st['row_loads'] / st['steps']:.3f} dir/step {st['dir_loads'] / st['steps']:.3f} "
        f"f_mis {st['jumps'] / st['steps']:.3f}")
if ncheck:
    import oracle

    hs = z["reads"][: ncheck * bp]
    ho = np.arange(ncheck + 1, dtype=np.uint64) * bp
    ds, do = oracle.digest_batch(oracle.DIGEST_PROMOTED, k, w, hs, ho)
    want = oracle.OracleIndex.from_raw(raw).pml(ds, do)
    got_offs = keep["oo"][: ncheck + 1].cpu().numpy().astype(np.uint64)
    same = bool(np.array_equal(got_offs, do) and np.array_equal(d_l16[: int(do[-1])].cpu().numpy().view(np.uint16).astype(np.uint32), want))
    line += f" | oracle gate on {ncheck} reads: {'equal' if same else 'DIFFERENT'}"
    assert same
print(line, flush=True)
