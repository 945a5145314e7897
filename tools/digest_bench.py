"""Throughput of the device digestion (spx_digest_batch_device) on the bench read shape:
N reads x 200 bp of random DNA, k=4, w=11, both kinds; the oracle's digestion timed beside it.

    python tools/digest_bench.py [--reads 10000000] [--len 200]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spumoni_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--len", type=int, default=200)
    ap.add_argument("--k", type=int, default=4)
    ap.add_argument("--w", type=int, default=11)
    ap.add_argument("--cpu-reads", type=int, default=200_000)
    ap.add_argument("--kernel", type=int, default=0, help="digest_kernel option: 0 automatic, 1 lane per read, 2 wavefront per read, 3 lane per chunk")
    a = ap.parse_args()
    ix = capi.digester(0)
    ix.set_option("digest_kernel", a.kernel)
    g = torch.Generator(device="cuda").manual_seed(1)
    n = a.reads * a.len
    d_seqs = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")[
        torch.randint(0, 4, (n + 64,), generator=g, device="cuda")]
    d_offs = torch.arange(a.reads + 1, dtype=torch.int64, device="cuda") * a.len
    # (the first tens of milliseconds of a process run at lower clocks: warm up, and take both kinds twice)
    for _ in range(10):
        ix.digest_device(capi.SPX_DIGEST_PROMOTED, a.k, a.w, d_seqs, d_offs, n)
    torch.cuda.synchronize()
    for kind, name in ((capi.SPX_DIGEST_PROMOTED, "-m promoted"), (capi.SPX_DIGEST_DNA, "-a dna"),
                       (capi.SPX_DIGEST_PROMOTED, "-m promoted"), (capi.SPX_DIGEST_DNA, "-a dna")):
        out, out_offs = ix.digest_device(kind, a.k, a.w, d_seqs, d_offs, n)  # warm-up (allocations)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        cap = out.numel()
        e0.record()
        for _ in range(reps):
            capi._check(capi.lib().spx_digest_batch_device(
                ix._h, kind, a.k, a.w, capi._t_ptr(d_seqs), capi._t_ptr(d_offs), a.reads, n, capi._t_ptr(out), cap,
                capi._t_ptr(out_offs), capi.C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        total = int(out_offs[-1].item())
        alg = 2 * n + total + 3 * 8 * (a.reads + 1)  # chars read by both passes, bytes written, offsets
        print(f"{name:12s} {a.reads} reads x {a.len}: {ms:8.3f} ms  {a.reads / ms / 1e3:8.1f} M reads/s  "
              f"{n / ms / 1e6:7.1f} G chars/s  algorithmic {alg / ms / 1e6:7.1f} GB/s  digested {total / n:.3f} B/char")
    # CPU: the oracle's digestion, one thread
    try:
        import oracle
        h = d_seqs[: a.cpu_reads * a.len].cpu().numpy()
        offs = (np.arange(a.cpu_reads + 1, dtype=np.uint64) * a.len)
        t0 = time.time()
        oracle.digest_batch(1, a.k, a.w, h, offs)
        dt = time.time() - t0
        print(f"oracle (1 thread) -m: {a.cpu_reads / dt / 1e6:.3f} M reads/s")
    except Exception as e:  # the oracle is optional here
        print("oracle not timed:", e)


if __name__ == "__main__":
    main()
