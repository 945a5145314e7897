"""End-to-end `spumoni run` (files in -> files out) on the 5-strain E. coli shape (run on the GPU box)."""
import os, subprocess, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spumoni_amd import synth
from tests.sdsl_files import write_null_db

d = "/tmp/e2e"; os.makedirs(d, exist_ok=True)
base = synth.random_genome(4_641_652, seed=1)
genomes = [base] + [synth.mutate(base, seed=s) for s in (2, 3, 4, 5)]
text, doc_lengths = synth.pangenome_text(genomes)
raw = synth.index_from_text(torch.from_numpy(text).cuda(), with_samples=False).cpu()
open(f"{d}/ref.fa", "w").write(">x\n")
raw.write_raw_files(f"{d}/ref.fa")
write_null_db(f"{d}/ref.fa.pmlnulldb", 3.0, [1, 2, 3, 3, 3, 3, 3])
nreads, m = int(os.environ.get("E2E_READS", "1000000")), 200
seqs, offs = synth.sample_reads(text, nreads, m, seed=12)
t0 = time.time()
with open(f"{d}/reads.fa", "wb") as f:
    rows = seqs.reshape(nreads, m)
    hdr = np.array([f">read_{i}\n".encode().ljust(16, b" ") for i in range(nreads)])  # fixed-width ids
    for i in range(0, nreads, 100000):
        blk = b"".join(b">read_%d\n" % j + rows[j].tobytes() + b"\n" for j in range(i, min(nreads, i + 100000)))
        f.write(blk)
print(f"wrote reads.fa ({os.path.getsize(d + '/reads.fa')/1e6:.0f} MB) in {time.time()-t0:.1f}s")
for rep in range(2):
    t0 = time.time()
    r = subprocess.run([f"{ROOT}/spumoni_amd/bin/spumoni", "run", "-r", f"{d}/ref", "-p", f"{d}/reads.fa", "-P", "-c", "-n"],
                       capture_output=True, env=dict(os.environ, SPUMONI_TIMING="1"))
    dt = time.time() - t0
    print([l for l in r.stderr.decode().splitlines() if "[timing]" in l])
    print(r.stderr.decode().replace("\033[32m", "").replace("\033[0m", "").strip().splitlines()[-4:])
    print(f"spumoni run -P -c -n: {dt:.2f}s wall for {nreads} reads = {nreads/dt/1e6:.2f} M reads/s; "
          f"pseudo_lengths {os.path.getsize(d + '/reads.fa.pseudo_lengths')/1e6:.0f} MB")
