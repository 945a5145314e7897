"""Times the CPU oracle at several thread counts on the bench index (run on the GPU box)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spumoni_amd import synth
import oracle

r = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 26
raw = synth.statistical_rlbwt(r, 253, 8.0, seed=3, device="cuda", zipf=1.0)
seqs, offs = synth.simulate_reads(raw, 400_000, 44, seed=13)
rawc = raw.cpu(); hs = seqs.cpu().numpy(); ho = offs.cpu().numpy()
t0 = time.time(); orc = oracle.OracleIndex.from_raw(rawc); print("oracle build", time.time() - t0, flush=True)
for T in (1, 8, 32, 64, 128, 256):
    nr = min(400_000, 3000 * T)
    t0 = time.time(); orc.pml(hs[: nr * 44], ho[: nr + 1], nthreads=T); dt = time.time() - t0
    print(f"threads {T:4d}: {nr} reads in {dt:.2f}s = {nr/dt:.0f} reads/s ({nr/dt/T:.0f}/thread)", flush=True)
print(open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0])
print("affinity", len(os.sched_getaffinity(0)))
try:
    print("cpu.max", open("/sys/fs/cgroup/cpu.max").read())
except Exception as e:
    print(e)

# tier T1 (Elias-Fano + wavelet tree + block walk: the reference's own structures)
t0 = time.time(); t1 = oracle.OracleT1Index.from_raw(rawc); print("T1 oracle build", time.time() - t0, flush=True)
for T in (1, 16, 32):
    nr = min(400_000, 1000 * T)
    t0 = time.time(); got = t1.pml(hs[: nr * 44], ho[: nr + 1], nthreads=T); dt = time.time() - t0
    print(f"T1 threads {T:4d}: {nr} reads in {dt:.2f}s = {nr/dt:.0f} reads/s ({nr/dt/T:.0f}/thread)", flush=True)
assert np.array_equal(got, orc.pml(hs[: nr * 44], ho[: nr + 1], nthreads=32))
print("T1 == T2 on the sample")
