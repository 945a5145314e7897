"""Stress of the C++ host harness WITHOUT a GPU (CPU tool): `spumoni run` -- plain, ASan + UBSan and TSan builds -- on 6 000
reads cut into 260 .. 1 040 super-batches (SPUMONI_SUPER_BATCH) dealt to 2 .. 5 workers, text from the boundary and
host-formatted, against tests/fake_device (the C-ABI answered by the CPU oracle; test infrastructure):

    python tools/host_stress.py        (builds tests/fake_device and the sanitizer builds of the host if they are not there)

Every run must exit 0 without a sanitizer report and write the oracle harness's bytes (profiles/r03_host_stress_cpu.txt)."""
import os, sys, subprocess, pathlib, shutil, tempfile, numpy as np, glob, filecmp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import test_gpu_cli as T

def fake_device_dir():
    """tests/fake_device built as libspumoni_gpu.so (FAKE_DEVICE_DIR: an existing build; otherwise built here, once)."""
    d = os.environ.get("FAKE_DEVICE_DIR") or os.path.join(tempfile.gettempdir(), "spumoni_fake_device")
    so = os.path.join(d, "libspumoni_gpu.so")
    src = [os.path.join(ROOT, "tests", "fake_device", "fake_spumoni_gpu.c"), os.path.join(ROOT, "oracle", "spumoni_oracle.c"),
           os.path.join(ROOT, "oracle", "orc_digest.c")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in src):
        os.makedirs(d, exist_ok=True)
        subprocess.check_call(["gcc", "-O1", "-g", "-std=c11", "-fsigned-char", "-fPIC", "-Wno-unknown-pragmas", "-shared", "-pthread", "-o", so] + src)
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "spumoni_amd", "csrc", "host"), "all", "san", "-j2"], stdout=subprocess.DEVNULL)
    return d

tmp = pathlib.Path('/tmp/stress'); shutil.rmtree(tmp, ignore_errors=True); tmp.mkdir()
fake = fake_device_dir()
ref, prefix, seqs, offs, rng = T._setup(tmp, 95, list(b"ACGT"), n=20000, nreads=6000)
T._write_fasta(tmp / "reads.fa", seqs, offs, np.random.default_rng(5))
o = subprocess.run([T.ORC_RUN, prefix, str(tmp / "reads.fa"), "P", "1", "1", "150", "n", prefix + ".rawtext"], capture_output=True); assert o.returncode == 0
for e in (".pseudo_lengths", ".doc_numbers", ".report"): shutil.move(str(tmp / "reads.fa") + e, str(tmp / ("want" + e)))
bad = 0
for rep in range(12):
    for which in ("tsan", "asan", ""):
        exe = os.path.join(ROOT, 'spumoni_amd', 'bin', 'spumoni') + ("_" + which if which else "")
        env = dict(os.environ, LD_LIBRARY_PATH=fake, SPUMONI_CACHE="off", SPUMONI_GPUS=",".join(["0"] * (2 + rep % 4)), SPUMONI_SUPER_BATCH=str(1000 + 700 * (rep % 5)),
                   TSAN_OPTIONS="report_signal_unsafe=0:history_size=4:exitcode=66", ASAN_OPTIONS="detect_leaks=0", SPUMONI_TEXT=prefix + ".rawtext")
        if rep % 3 == 2: env["SPUMONI_HOST_FORMAT"] = "1"
        pre = ["setarch", "x86_64", "-R"] if which == "tsan" else []
        r = subprocess.run(pre + [exe, "run", "-r", ref, "-p", str(tmp / "reads.fa"), "-n", "-P", "-c", "-d"], capture_output=True, env=env)
        ok = r.returncode == 0 and not any(b in r.stderr for b in (b"Sanitizer", b"runtime error"))
        same = all(filecmp.cmp(str(tmp / "reads.fa") + e, str(tmp / ("want" + e)), shallow=False) for e in (".pseudo_lengths", ".doc_numbers", ".report"))
        nb = sum(int(l.split("(")[1].split()[0]) for l in r.stderr.decode(errors='replace').splitlines() if "super-batches" in l)
        if not (ok and same):
            bad += 1
            print("PROBLEM", rep, which, r.returncode, same, r.stderr.decode(errors='replace')[-1500:])
        else:
            print(rep, which or "plain", "ok", nb, "super-batches")
print("bad", bad)
