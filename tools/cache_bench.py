"""Start-up of an index (run on the GPU box): raw run files -> parse + flatten (spx_index_load_raw) against the
flat-layout cache (spx_index_save / spx_index_load_flat), each in a FRESH process (releasing a 50-200 GB index
is charged to whatever allocates next in the same process), and a device-to-device clone.  Files live on tmpfs.
usage: cache_bench.py [runs]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
d = "/dev/shm/cache_bench"

def child(what):
    import torch
    from spumoni_amd import capi
    torch.cuda.init()
    t0 = time.time()
    if what == "raw":
        ix = capi.Index.load_raw(f"{d}/ref", capi.SPX_MODE_PML, 0)
        t = time.time() - t0
        desc = ix.describe()
        t0 = time.time(); ix.save(f"{d}/ref.pml.spx"); ts = time.time() - t0
        tc = None
        if desc["device_bytes"] < 120e9:
            t0 = time.time(); dup = ix.clone(0); torch.cuda.synchronize(); tc = time.time() - t0
        print("RESULT", t, ts, tc, desc["device_bytes"], desc["fat_slots_per_run"])
    else:
        ix = capi.Index.load_flat(f"{d}/ref.pml.spx", 0)
        print("RESULT", time.time() - t0)
    sys.stdout.flush()
    os._exit(0)  # skip the teardown of a 50-200 GB index

if len(sys.argv) > 2 and sys.argv[1] == "--child":
    child(sys.argv[2])

r = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 27
os.makedirs(d, exist_ok=True)
import torch
from spumoni_amd import synth
raw = synth.statistical_rlbwt(r, 253, 8.0, seed=3, device="cuda", zipf=1.0)
raw.cpu().write_raw_files(f"{d}/ref")
del raw; torch.cuda.empty_cache()
raw_bytes = sum(os.path.getsize(f"{d}/ref.{e}") for e in ("bwt.heads", "bwt.len", "thr_pos"))

def run(what):
    o = subprocess.run([sys.executable, __file__, "--child", what], capture_output=True, text=True)
    line = [l for l in o.stdout.splitlines() if l.startswith("RESULT")]
    assert line, o.stdout + o.stderr
    return [None if x == "None" else float(x) for x in line[0].split()[1:]]

t_raw, t_save, t_clone, dev_bytes, spr = run("raw")
sz = os.path.getsize(f"{d}/ref.pml.spx")
t_load = min(run("flat")[0] for _ in range(2))
print(f"r = {r}: raw files {raw_bytes/1e9:.2f} GB; flat index {dev_bytes/1e9:.1f} GB on the device, {spr:.1f} fat slots per run; "
      f"cache file {sz/1e9:.1f} GB (the fat table is not stored: rebuilt on the device)")
print(f"  spx_index_load_raw  (read + unpack 5-byte records on the host, copy, flatten on the GPU): {t_raw:.2f} s")
print(f"  spx_index_save      : {t_save:.2f} s = {sz/1e9/t_save:.1f} GB/s")
print(f"  spx_index_load_flat (file -> page-locked staging -> device, 4 readers; fat table rebuilt): {t_load:.2f} s = {sz/1e9/t_load:.1f} GB/s of file")
if t_clone:
    print(f"  spx_index_clone     (device to device, same device here): {t_clone:.2f} s = {dev_bytes/1e9/t_clone:.0f} GB/s")
for f in os.listdir(d):
    os.remove(os.path.join(d, f))
