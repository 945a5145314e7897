"""Mutation fuzz of the serialised-index reader (spumoni_amd/csrc/host/index_files.cpp: the sdsl-lite / r-index streams of
<ref>.thrbv.spumoni) on the CPU: a valid file written by the tests' writer, then truncated, with a few bytes flipped,
with an aligned word overwritten by a wild value (0, 2^31, 2^40, 2^63, 2^64 - 1, the file's size in bits ...), or with
its first 200 bytes disturbed -- through `spumoni_asan dump-index`: every file is either decoded or refused with a message,
never a sanitizer report, an abort or a hang.  (The first run found size fields that sized allocations before they were
held against the file's length: 2^60-byte vectors.)

    python tools/index_reader_fuzz.py [mutations]
"""
import os, sys, subprocess, pathlib, shutil, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import cases, sdsl_files

tmp = pathlib.Path(os.environ.get("INDEX_FUZZ_DIR") or tempfile.mkdtemp(prefix="idx_fuzz_"))
tmp.mkdir(parents=True, exist_ok=True)
HOST = os.path.join(ROOT, "spumoni_amd", "bin", "spumoni_asan")
raw, _ = cases.real_case(7, 3000, list(b"ACGT"), ndocs=2)
good = str(tmp / "good.thrbv.spumoni")
sdsl_files.write_thrbv(good, raw.heads.numpy(), raw.lens.numpy(), raw.thr.numpy())
blob = bytearray(open(good, "rb").read())
env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:max_allocation_size_mb=4096:allocator_may_return_null=1")
assert subprocess.run([HOST, "dump-index", good, "P"], capture_output=True, env=env).returncode == 0
WILD = [0, 1, 2**31, 2**40, 2**63, 2**64 - 1, len(blob) * 8, len(blob) * 8 + 64]
bad = accepted = refused = 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for seed in range(N):
    rng = np.random.default_rng(seed)
    b = bytearray(blob)
    kind = seed % 4
    if kind == 0:
        b = b[: int(rng.integers(0, len(b)))]
    elif kind == 1:
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
    elif kind == 2:
        at = int(rng.integers(0, len(b) // 8)) * 8
        b[at:at + 8] = WILD[int(rng.integers(0, len(WILD)))].to_bytes(8, "little")
    else:
        for _ in range(int(rng.integers(1, 3))):
            b[int(rng.integers(0, min(200, len(b))))] = int(rng.integers(0, 256))
    p = str(tmp / "m.thrbv.spumoni")
    open(p, "wb").write(bytes(b))
    try:
        r = subprocess.run([HOST, "dump-index", p, "P"], capture_output=True, env=env, timeout=60)
    except subprocess.TimeoutExpired:
        bad += 1
        print("seed", seed, "kind", kind, "TIMEOUT")
        continue
    if b"Sanitizer" in r.stderr or b"runtime error" in r.stderr or r.returncode < 0:
        bad += 1
        print("seed", seed, "kind", kind, "rc", r.returncode, r.stderr.decode(errors="replace")[:300])
    elif r.returncode == 0:
        accepted += 1
    else:
        refused += 1
print("mutations", N, "bad", bad, "decoded", accepted, "refused", refused)
sys.exit(1 if bad else 0)
