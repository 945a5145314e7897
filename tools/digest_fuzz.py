"""Random batches through every digestion kernel (automatic / lane per read / wavefront per read / lane per chunk) against the
oracle: lengths from a few shapes (short, around the chunk size, long), characters outside ACGT at a random rate.
   python tools/digest_fuzz.py [batches = 100] [first seed = 0]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from spumoni_amd import capi

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 100
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ix = capi.digester(0)
bad = 0
for seed in range(first, first + nb):
    rng = np.random.default_rng(seed)
    shape = seed % 4
    nreads = int(rng.integers(1, 300))
    if shape == 0:
        lens = rng.integers(0, 400, size=nreads)
    elif shape == 1:
        lens = rng.choice([239, 240, 241, 251, 252, 253, 479, 480, 481, 0, 10, 11, 12], size=nreads)
    elif shape == 2:
        lens = rng.integers(600, 6000, size=nreads)
    else:
        lens = np.where(rng.random(nreads) < 0.2, rng.integers(3000, 30000, size=nreads), rng.integers(0, 900, size=nreads))
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    seqs = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=int(offs[-1]))].copy()
    p_n = float(rng.choice([0.0, 0.0, 0.0005, 0.01, 0.2]))
    if p_n:
        seqs[rng.random(seqs.size) < p_n] = rng.choice([ord("N"), ord("a"), 0, 255])
    for kind in (1, 2):
        want, want_offs = oracle.digest_batch(kind, 4, 11, seqs, offs)
        for forced in (0, 1, 2, 3):
            ix.set_option("digest_kernel", forced)
            got, got_offs = ix.digest_host(kind, 4, 11, seqs, offs)
            if not (np.array_equal(got_offs, want_offs) and np.array_equal(got, want)):
                bad += 1
                print(f"MISMATCH seed {seed} kind {kind} kernel {forced} shape {shape} p_n {p_n}", flush=True)
print(f"{nb} batches from seed {first}: {bad} mismatches")
sys.exit(1 if bad else 0)
