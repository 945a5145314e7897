"""Differential fuzz of the C++ host harness WITHOUT a GPU (CPU tool): random FASTA / FASTQ files -- descriptions and tabs in
the headers, headers without an id, multi-line and empty reads, lower case, N, CR LF, trailing blanks, blank lines, no final
newline -- through `spumoni run` (ASan + UBSan build, 1-3 workers, super-batches of 1 000 / 2 500 / 10^7 characters, text from
the boundary or formatted on the host, PML / MS, with and without documents and reports) against tests/fake_device, and through
the oracle harness (oracle/orc_run): the same exit status, the same error message, the same bytes in every output file --
also in the files a run that ends in a fatal error leaves behind.

    python tools/cli_fuzz_cpu.py [seeds [first seed]]        (builds tests/fake_device and the sanitizer builds of the host if
                                                              they are not there; FAKE_DEVICE_DIR: an existing build;
                                                              CLI_FUZZ_BIN=spumoni_tsan | spumoni: another build of the host)
"""
import os, sys, subprocess, pathlib, shutil, tempfile, numpy as np, filecmp
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

FAKE = fake_device_dir()
tmp = pathlib.Path(os.environ.get('CLI_FUZZ_DIR', '/tmp/cli_fuzz')); shutil.rmtree(tmp, ignore_errors=True); tmp.mkdir(parents=True)
ref, prefix, seqs, offs, rng0 = T._setup(tmp, 97, list(b"ACGT"), n=8000, nreads=10)
text = np.fromfile(prefix + ".rawtext", dtype=np.uint8)
HOST = os.path.join(ROOT, 'spumoni_amd', 'bin', os.environ.get('CLI_FUZZ_BIN', 'spumoni_asan'))  # spumoni_tsan: races instead of addresses
PRE = ["setarch", "x86_64", "-R"] if HOST.endswith("_tsan") and shutil.which("setarch") else []  # (gcc 11's TSan wants ASLR off here)

def rand_seq(rng, n):
    if n >= text.size - 1:
        s = np.tile(text, n // text.size + 1)[:n].copy()   # (longer than the text: the text over and over)
    elif rng.random() < 0.6:
        a = int(rng.integers(0, text.size - n - 1)); s = text[a:a + n].copy()
    else:
        s = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=n)
    if rng.random() < 0.3: s = np.frombuffer(s.tobytes().lower(), dtype=np.uint8)
    return s.tobytes()

def make_file(rng, hostile):
    fq = rng.random() < 0.35
    out = []
    nrec = int(rng.integers(0, 40))
    for i in range(nrec):
        name = b"r%d" % i
        u = rng.random()
        if u < 0.15: name += b" some description"
        elif u < 0.25: name += b"\tx y"
        elif u < 0.25 + 0.05 * hostile: name = b"x" * int(rng.integers(1, 3))   # short header (<= 2 chars with the mark -> fatal)
        elif u < 0.25 + 0.08 * hostile: name = b""                               # bare mark
        n = int(rng.choice([0, 1, 3, 20, 80, 150, 400, 1200], p=[0.04 * hostile, 0.06, 0.1, 0.2, 0.25 + 0.04 * (1 - hostile), 0.2, 0.1, 0.05]))
        if rng.random() < 0.004: n = int(rng.integers(65530, 66000))   # a read that does not fit 16-bit values: the batch goes the 32-bit way
        if rng.random() < 0.01: name += b" " + b"d" * int(rng.integers(200, 6000))   # a very long header line
        s = rand_seq(rng, n) if n else b""
        eol = b"\r\n" if rng.random() < 0.1 else b"\n"
        trail = b"  " if rng.random() < 0.1 else b""
        if fq:
            out.append(b"@" + name + eol + s + trail + eol + b"+" + (name if rng.random() < 0.2 else b"") + eol + b"I" * len(s) + eol)
        else:
            lines = []
            if n and rng.random() < 0.5:
                cut = int(rng.integers(1, n + 1)); lines = [s[:cut], s[cut:]] if cut < n else [s]
            else:
                lines = [s]
            rec = b">" + name + eol + b"".join(l + trail + eol for l in lines)
            if rng.random() < 0.05: rec += eol        # blank line after the record
            out.append(rec)
    data = b"".join(out)
    if data and rng.random() < 0.3: data = data.rstrip(b"\r\n")
    return data

bad = 0; fatals = 0; empties = 0; total_reads = 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for seed in range(first, first + N):
    rng = np.random.default_rng(seed)
    hostile = seed % 4 == 0   # one file in four with headers without an id and empty reads: runs that end in a fatal error
    data = make_file(rng, 1.0 if hostile else 0.0)
    mode = "P" if rng.random() < 0.6 else "M"
    doc = int(rng.random() < 0.5); rep = int(rng.random() < 0.7)
    for d in ("cli", "orc"):
        shutil.rmtree(tmp / d, ignore_errors=True); (tmp / d).mkdir()
        (tmp / d / "reads.fa").write_bytes(data)
    env = dict(os.environ, LD_LIBRARY_PATH=FAKE, SPUMONI_CACHE="off", SPUMONI_TEXT=prefix + ".rawtext", ASAN_OPTIONS="detect_leaks=0", TSAN_OPTIONS="report_signal_unsafe=0:history_size=4",
               SPUMONI_SUPER_BATCH=str(int(rng.choice([1000, 2500, 10**7]))), SPUMONI_GPUS=",".join(["0"] * int(rng.integers(1, 4))))
    if rng.random() < 0.3: env["SPUMONI_HOST_FORMAT"] = "1"
    # the classifier's inputs: bin width (-w, 50 .. 400 or refused) and the null database's percentile (-> max_value_thr,
    # compute_ms_pml.cpp:864-875 / 1054-1063)
    bw = int(rng.choice([150, 50, 400, 77, 233, 49, 401]))
    from tests.sdsl_files import write_null_db
    pct = float(rng.choice([0.0, 1.0, 3.0, 4.0, 9.0, 17.5, 60.0, 250.0]))
    for pf in {prefix, str(tmp / "promoted" / "ref.bin")}:
        if os.path.exists(pf):
            write_null_db(pf + ".pmlnulldb", pct, [1, 2, 3, 4]); write_null_db(pf + ".msnulldb", pct + 5, [5, 9, 9])
    flags = (["-c"] if rep else []) + (["-d"] if doc else []) + (["-w", str(bw)] if bw != 150 else [])
    if rng.random() < 0.5: flags += ["-t", str(int(rng.integers(1, 9)))]  # (-t: the host threads that parse and format)
    # digestion (run -m / -a, compute_ms_pml.cpp:919-931): which index the digested reads are searched in does not matter
    # to the host; -m wants the index under <ref>.bin
    digest = str(rng.choice(["n", "n", "m", "a"]))
    kw, orc_kw, pfx, ref_run = [], [], prefix, ref
    if digest != "n":
        k = int(rng.integers(1, 5)); w = k + int(rng.integers(0, 9))
        kw, orc_kw = ["-K", str(k), "-W", str(w)], ["--k", str(k), "--w", str(w)]
        if digest == "m":  # (a directory of its own: <ref>.fa and <ref>.bin side by side are refused)
            ref_run = str(tmp / "promoted" / "ref"); pfx = ref_run + ".bin"
            if not os.path.exists(pfx):
                (tmp / "promoted").mkdir(exist_ok=True)
                for f in os.listdir(tmp):
                    if f.startswith("ref.fa"):
                        shutil.copy(tmp / f, tmp / "promoted" / ("ref.bin" + f[len("ref.fa"):]))
    r = subprocess.run(PRE + [HOST, "run", "-r", ref_run, "-p", str(tmp / "cli" / "reads.fa"), "-" + digest, "-" + mode] + flags + kw, capture_output=True, env=env)
    o = subprocess.run([T.ORC_RUN, pfx, str(tmp / "orc" / "reads.fa"), mode, str(doc), str(rep), str(bw), digest, prefix + ".rawtext"] + orc_kw, capture_output=True)
    problems = []; fatals += o.returncode != 0; empties += (len(data) == 0)
    if bw < 50 or bw > 400:  # include/spumoni_main.hpp:318-320: refused before anything is read
        if r.returncode != 1 or b"bin size used is not optimal" not in r.stderr:
            bad += 1; print("seed", seed, "bin width", bw, "was not refused", r.returncode)
        continue
    if b"Sanitizer" in r.stderr or b"runtime error" in r.stderr: problems.append("sanitizer")
    if (r.returncode == 0) != (o.returncode == 0): problems.append(f"rc {r.returncode} vs {o.returncode}")
    if o.returncode != 0:
        import re
        em = re.findall(rb"Error: \x1b\[0m(.*)", o.stderr); rm = re.findall(rb"Error: \x1b\[0m(.*)", r.stderr)
        if em != rm: problems.append(f"messages {rm} vs {em}")
    exts = ([".pseudo_lengths"] if mode == "P" else [".lengths", ".pointers"]) + ([".doc_numbers"] if doc else []) + ([".report"] if rep else [])
    for e in exts:
        a, b = tmp / "cli" / ("reads.fa" + e), tmp / "orc" / ("reads.fa" + e)
        ea, eb = a.exists(), b.exists()
        if ea != eb: problems.append(f"{e} exists {ea} vs {eb}")
        elif ea and not filecmp.cmp(str(a), str(b), shallow=False): problems.append(f"{e} differs ({a.stat().st_size} vs {b.stat().st_size})")
    if problems:
        bad += 1
        print("seed", seed, mode, digest, kw, flags, {k: env.get(k) for k in ("SPUMONI_SUPER_BATCH", "SPUMONI_GPUS", "SPUMONI_HOST_FORMAT")}, problems)
        shutil.copy(tmp / "cli" / "reads.fa", tmp / f"bad_{seed}.fa")
        if bad > 12: break
print("seeds", first, "..", first + N - 1, "bad", bad, "runs that ended in a fatal error", fatals, "empty files", empties)
sys.exit(1 if bad else 0)
