"""host_path: the PCIe-inclusive call spx_query_batch16 from page-locked buffers, timed call by call in ONE process
(27 ms in some processes, 44 ms in others).  Run several processes / environments from tools/host_path_modes.sh."""
import os, sys, time
import ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spumoni_amd import capi, synth

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 26
raw = synth.statistical_rlbwt(runs, 253, 8.0, seed=3, device="cuda", zipf=1.0)
seqs, offs = synth.simulate_reads(raw, 10_000_000, 44, seed=13, warmup=4)
ix = capi.Index.from_raw(raw, 0)
if not os.environ.get("HP_KEEP_RAW"):  # (freed device memory is wiped by the driver, in the background, with the copy engines)
    del raw; torch.cuda.empty_cache()
hs, ho = seqs.cpu().numpy(), offs.cpu().numpy()
tot, nreads = hs.size, ho.size - 1
ps, o1 = capi.pinned_array((tot,), np.uint8); ps[:] = hs
po, o2 = capi.pinned_array((nreads + 1,), np.uint64); po[:] = ho
pl16, o5 = capi.pinned_array((tot + 8,), np.uint16)
pc, o4 = capi.pinned_array((nreads,), capi.CLASS_DTYPE)
vp = lambda a: a.ctypes.data_as(C.c_void_p)
ms, kms = [], []
if os.environ.get("HP_CPU_SPIN"):  # the host core busy, the GPU idle
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < float(os.environ["HP_CPU_SPIN"]):
        pass
if os.environ.get("HP_GPU_ONLY"):  # the GPU busy, the host thread asleep in one synchronize
    x = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
    for _ in range(int(400 * float(os.environ["HP_GPU_ONLY"]))): x.mul_(1.0001)
    torch.cuda.synchronize()
if os.environ.get("HP_SPIN"):  # keep the compute units busy first: are the slow calls a clock ramp?
    x = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < float(os.environ["HP_SPIN"]):
        for _ in range(20): x.mul_(1.0001)
        torch.cuda.synchronize()
for rep in range(int(os.environ.get("HP_REPS", "12"))):
    t0 = time.perf_counter()
    rc = capi.lib().spx_query_batch16(ix._h, capi.SPX_MODE_PML, vp(ps), vp(po), nreads, vp(pl16), None, None, vp(pc), 150, 5)
    ms.append((time.perf_counter() - t0) * 1e3)
    assert rc == 0
    kms.append(ix.last_stats()["kernel_ms"])
st = ix.last_stats()
print(f"{os.environ.get('HP_TAG', 'default'):28s} kernel {st['kernel_ms']:6.2f} ms | per call: " + " ".join(f"{x:5.1f}" for x in ms)
      + " | walk events first..last: " + " ".join(f"{x:4.1f}" for x in kms)
      + f" | median {np.median(ms[2:]):5.1f} ms = {1e7 / np.median(ms[2:]) / 1e3:6.1f} M reads/s", flush=True)
