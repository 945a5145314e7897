"""Is the slow start of the host path ours?  A bare copy loop: page-locked host <-> device copies of 128 MB, back to
back from a cold start, GB/s per copy.  (host_path ran 27 ms in some processes and 44 ms in others.)"""
import time, torch
n = 128 << 20
h = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for name, (src, dst) in (("D2H", (d, h)), ("H2D", (h, d))):
    time.sleep(2.0)  # let the device go idle again
    t00 = time.perf_counter()
    out = []
    for i in range(60):
        t0 = time.perf_counter()
        dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out.append((time.perf_counter() - t00, n / dt / 1e9))
    print(name, "GB/s by copy (elapsed s: GB/s):", " ".join(f"{t:.2f}:{g:.0f}" for t, g in out[::3]))
# the same while a kernel keeps the compute units busy (does compute activity wake the copy path?)
x = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
time.sleep(2.0)
for i in range(50): x.mul_(1.0001)
torch.cuda.synchronize()
t0 = time.perf_counter(); h.copy_(d, non_blocking=True); torch.cuda.synchronize()
print(f"D2H right after 50 compute kernels: {n / (time.perf_counter() - t0) / 1e9:.0f} GB/s")
