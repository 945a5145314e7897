"""End-to-end `spumoni run` (files in -> files out, on tmpfs) on the 5-strain E. coli shape.

Run on the GPU box (gpurun).  Prints, for E2E_READS reads of 200 bp:
  * our CLI: first run (raw index files -> flatten, SPUMONI_CACHE=write), second run (flat-layout
    cache), SPUMONI_GPUS=0,0 (two workers, one queue), SPUMONI_REPORT_ONLY=1;
  * the CPU oracle harness (oracle/orc_run, single thread, the reference's loop shape) on the first
    E2E_CPU_READS reads: the file-emission-inclusive CPU baseline SURVEY 8(d) asks for;
  * cmp of every output file of the two on that sample.
"""
import os, subprocess, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spumoni_amd import synth
from tests.sdsl_files import write_null_db

d = os.environ.get("E2E_DIR", "/dev/shm/e2e"); os.makedirs(d, exist_ok=True)
base = synth.random_genome(4_641_652, seed=1)
genomes = [base] + [synth.mutate(base, seed=s) for s in (2, 3, 4, 5)]
text, doc_lengths = synth.pangenome_text(genomes)
raw = synth.index_from_text(torch.from_numpy(text).cuda(), with_samples=False).cpu()
open(f"{d}/ref.fa", "w").write(">x\n")
raw.write_raw_files(f"{d}/ref.fa")
write_null_db(f"{d}/ref.fa.pmlnulldb", 3.0, [1, 2, 3, 3, 3, 3, 3])
nreads, m = int(os.environ.get("E2E_READS", "4000000")), 200
ncpu = int(os.environ.get("E2E_CPU_READS", "50000"))
seqs, offs = synth.sample_reads(text, nreads, m, seed=12, null_fraction=float(os.environ.get("E2E_NULL_FRACTION", "0.5")))


def write_fasta(path, lo, hi):
    rows = seqs.reshape(nreads, m)
    with open(path, "wb") as f:
        for i in range(lo, hi, 100000):
            f.write(b"".join(b">read_%d\n" % j + rows[j].tobytes() + b"\n" for j in range(i, min(hi, i + 100000))))


t0 = time.time()
write_fasta(f"{d}/reads.fa", 0, nreads)
write_fasta(f"{d}/sample.fa", 0, ncpu)
print(f"wrote reads.fa ({os.path.getsize(d + '/reads.fa')/1e6:.0f} MB) in {time.time()-t0:.1f}s", flush=True)
for f in os.listdir(d):
    if f.endswith(".spx"):
        os.remove(os.path.join(d, f))


if os.environ.get("E2E_ONLY_SETUP"):  # (tools/first_8gpu.sh: the files, then its own runs)
    print("setup only:", d)
    sys.exit(0)


def run(tag, reads, extra_env):
    env = dict(os.environ, **extra_env)
    t0 = time.time()
    extra = extra_env.pop("E2E_EXTRA", "").split()
    env = dict(os.environ, **extra_env)
    r = subprocess.run([f"{ROOT}/spumoni_amd/bin/spumoni", "run", "-r", f"{d}/ref", "-p", reads, "-P", "-c", "-n"] + extra,
                       capture_output=True, env=env)
    dt = time.time() - t0
    err = r.stderr.decode().replace("\033[32m", "").replace("\033[0m", "")
    assert r.returncode == 0, err
    n = nreads if reads.endswith("reads.fa") else ncpu
    import re
    secs = [float(x) for x in re.findall(r"done\.\s+\(([0-9.]+) sec\)", err)]  # index load, processing the patterns
    load_s, proc_s = (secs + [0, 0])[:2]
    print(f"== {tag}: {dt:.2f}s wall (process start to exit), {n/dt/1e6:.2f} M reads/s end to end; "
          f"loading the index {load_s:.3f}s, processing the patterns {proc_s:.3f}s = {n/max(proc_s,1e-9)/1e6:.2f} M reads/s "
          f"(pseudo_lengths {os.path.getsize(reads + '.pseudo_lengths')/1e6:.0f} MB)")
    for l in err.splitlines():
        if "[timing]" in l:
            print("   ", l.strip())
    sys.stdout.flush()
    return dt


run("raw index files, SPUMONI_CACHE=write", f"{d}/reads.fa", {"SPUMONI_CACHE": "write"})
run("flat-layout cache (default: SPUMONI_GPUS=0,0,0 -- three workers, one copy of the index)", f"{d}/reads.fa", {})
if os.environ.get("E2E_PREP_AB"):  # round 6: how the output files' pages are had (classify.cpp, prepare_one), interleaved
    modes = os.environ["E2E_PREP_AB"].split(",") if "," in os.environ["E2E_PREP_AB"] else ["falloc", "populate:4", "populate:16", "populate:64"]
    what = {"falloc": "the default: fallocate, THEN 4 threads make the page table entries", "populate": "ftruncate; MADV_POPULATE_WRITE makes pages and entries"}
    for rep in range(int(os.environ.get("E2E_PREP_REPS", "2"))):
        for md in modes:
            name, _, nt = md.partition(":")
            env = {"SPUMONI_PREP": name}
            if nt:
                env["SPUMONI_PREP_THREADS"] = nt
            run(f"SPUMONI_PREP={md} ({what[name]})", f"{d}/reads.fa", env)
    sys.exit(0)
quick = os.environ.get("E2E_QUICK") is not None  # (only the first two runs of the PML block)
run("flat-layout cache, again", f"{d}/reads.fa", {})
if not quick:
    run("flat-layout cache, SPUMONI_HOST_FORMAT=1 (round 2: values over PCIe, digits on the host)", f"{d}/reads.fa", {"SPUMONI_HOST_FORMAT": "1"})
    os.replace(f"{d}/reads.fa.pseudo_lengths", f"{d}/host_format.pseudo_lengths")
    run("flat-layout cache, text from the device again", f"{d}/reads.fa", {})
    print("   cmp host-formatted against device-formatted .pseudo_lengths:",
          "identical" if subprocess.run(["cmp", f"{d}/reads.fa.pseudo_lengths", f"{d}/host_format.pseudo_lengths"]).returncode == 0 else "DIFFERENT", flush=True)
    os.remove(f"{d}/host_format.pseudo_lengths")
    run("SPUMONI_MAP_OUTPUT=0 (plain writes: one pwrite stream per file)", f"{d}/reads.fa", {"SPUMONI_MAP_OUTPUT": "0"})
    run("SPUMONI_MAP_OUTPUT=nopin (tails mapped, not registered: the pool copies the text in)", f"{d}/reads.fa", {"SPUMONI_MAP_OUTPUT": "nopin"})
    run("SPUMONI_MAP_FACTOR=0.5 (the estimate is short: half the file goes through the writer thread)", f"{d}/reads.fa", {"SPUMONI_MAP_FACTOR": "0.5"})
    run("SPUMONI_PIN_SHARE=1 (the whole tail registered with the device, the estimate's excess cut at the end: the round's first form)", f"{d}/reads.fa", {"SPUMONI_PIN_SHARE": "1"})
    run("the default again", f"{d}/reads.fa", {})
    run("SPUMONI_GPUS=0 (one worker)", f"{d}/reads.fa", {"SPUMONI_GPUS": "0"})
    run("SPUMONI_GPUS=0,0 (two workers on one device)", f"{d}/reads.fa", {"SPUMONI_GPUS": "0,0"})
    for mb in (8, 16, 32, 128):
        run(f"SPUMONI_SUPER_BATCH={mb} MB", f"{d}/reads.fa", {"SPUMONI_SUPER_BATCH": str(mb << 20)})
    run("SPUMONI_REPORT_ONLY=1", f"{d}/reads.fa", {"SPUMONI_REPORT_ONLY": "1"})
    run("SPUMONI_REPORT_ONLY=1, again", f"{d}/reads.fa", {"SPUMONI_REPORT_ONLY": "1"})
    run("SPUMONI_REPORT_ONLY=1 SPUMONI_GPUS=0 (one worker)", f"{d}/reads.fa", {"SPUMONI_REPORT_ONLY": "1", "SPUMONI_GPUS": "0"})
    for mb in (8, 16, 32):
        run(f"SPUMONI_REPORT_ONLY=1 SPUMONI_SUPER_BATCH={mb} MB", f"{d}/reads.fa", {"SPUMONI_REPORT_ONLY": "1", "SPUMONI_SUPER_BATCH": str(mb << 20)})
    run("-t 8 (a pool of eight)", f"{d}/reads.fa", {"E2E_EXTRA": "-t 8"})
    run("-t 4 (a pool of four)", f"{d}/reads.fa", {"E2E_EXTRA": "-t 4"})
    run("-t 4, report only", f"{d}/reads.fa", {"E2E_EXTRA": "-t 4", "SPUMONI_REPORT_ONLY": "1"})
    run("-t 2, report only", f"{d}/reads.fa", {"E2E_EXTRA": "-t 2", "SPUMONI_REPORT_ONLY": "1"})
# ---- FASTQ input: the same reads with qualities (the output tails are sized for half the file: batch_loader.cpp:30-38) ----
if os.environ.get("E2E_FASTQ", "1") != "0" and not quick:
    nq = min(nreads, 2000000)
    rows = seqs.reshape(nreads, m)
    qual = b"I" * m
    write_fasta(f"{d}/half.fa", 0, nq)
    with open(f"{d}/halfq.fa", "wb") as f:
        for i in range(0, nq, 100000):
            f.write(b"".join(b"@read_%d\n" % j + rows[j].tobytes() + b"\n+\n" + qual + b"\n" for j in range(i, min(nq, i + 100000))))
    saved = nreads
    nreads_for_rate = nq

    def run_q(tag, reads):
        t0 = time.time()
        r = subprocess.run([f"{ROOT}/spumoni_amd/bin/spumoni", "run", "-r", f"{d}/ref", "-p", reads, "-P", "-c", "-n"], capture_output=True)
        dt = time.time() - t0
        err = r.stderr.decode().replace("\033[32m", "").replace("\033[0m", "")
        assert r.returncode == 0, err
        import re
        secs = [float(x) for x in re.findall(r"done\.\s+\(([0-9.]+) sec\)", err)]
        print(f"== {tag}: {dt:.2f}s wall; loading the index {secs[0]:.3f}s, processing the patterns {secs[1]:.3f}s = {nq/max(secs[1],1e-9)/1e6:.2f} M reads/s")
        for l in err.splitlines():
            if "first super-batch" in l or "output bytes" in l:
                print("   ", l.strip())
        sys.stdout.flush()

    run_q(f"FASTA, {nq} reads", f"{d}/half.fa")
    run_q(f"FASTQ content (in a file named .fa: the reference accepts the content, not the extension), the same {nq} reads with qualities", f"{d}/halfq.fa")
    for e in (".pseudo_lengths", ".report"):
        same = subprocess.run(["cmp", f"{d}/half.fa{e}", f"{d}/halfq.fa{e}"]).returncode == 0
        print(f"   cmp {e} (FASTA against FASTQ input): {'identical' if same else 'DIFFERENT'}", flush=True)
    for f_ in ("half.fa", "halfq.fa"):
        for e in ("", ".pseudo_lengths", ".report"):
            if os.path.exists(f"{d}/{f_}{e}"):
                os.remove(f"{d}/{f_}{e}")
# ---- MS mode: three output files side by side (lengths, pointers, report) ----
if os.environ.get("E2E_MS", "1") != "0":
    dm = d + "/ms"; os.makedirs(dm, exist_ok=True)
    raw_ms = synth.index_from_text(torch.from_numpy(text).cuda(), with_samples=True).cpu()
    open(f"{dm}/ref.fa", "w").write(">x\n")
    raw_ms.write_raw_files(f"{dm}/ref.fa")
    text.tofile(f"{dm}/ref.fa.rawtext")
    write_null_db(f"{dm}/ref.fa.msnulldb", 8.0, [5, 9, 9, 9])
    n_ms = min(nreads, int(os.environ.get("E2E_MS_READS", "1000000")))
    write_fasta(f"{dm}/reads.fa", 0, n_ms)

    def run_ms(tag, extra_env):
        env = dict(os.environ, SPUMONI_TEXT=f"{dm}/ref.fa.rawtext", **extra_env)
        t0 = time.time()
        r = subprocess.run([f"{ROOT}/spumoni_amd/bin/spumoni", "run", "-r", f"{dm}/ref", "-p", f"{dm}/reads.fa", "-M", "-c", "-n"], capture_output=True, env=env)
        dt = time.time() - t0
        err = r.stderr.decode().replace("\033[32m", "").replace("\033[0m", "")
        assert r.returncode == 0, err
        import re
        secs = [float(x) for x in re.findall(r"done\.\s+\(([0-9.]+) sec\)", err)]
        load_s, proc_s = (secs + [0, 0])[:2]
        sizes = {e: os.path.getsize(f"{dm}/reads.fa" + e) / 1e6 for e in (".lengths", ".pointers", ".report")}
        print(f"== -M -c -n, {n_ms} reads, {tag}: {dt:.2f}s wall; loading the index {load_s:.3f}s, processing the reads {proc_s:.3f}s = "
              f"{n_ms / max(proc_s, 1e-9) / 1e6:.2f} M reads/s (lengths {sizes['.lengths']:.0f} MB, pointers {sizes['.pointers']:.0f} MB, report {sizes['.report']:.0f} MB)")
        for l in err.splitlines():
            if "[timing]" in l or "[calls]" in l or "[phases]" in l:
                print("   ", l.strip())
        sys.stdout.flush()

    run_ms("SPUMONI_CACHE=write", {"SPUMONI_CACHE": "write"})
    run_ms("flat-layout cache", {})
    if os.environ.get("E2E_MS_TRACE"):  # (when every worker was inside the library, and the device's own phase times)
        run_ms("flat-layout cache, SPUMONI_CALL_TRACE + SPX_PHASE_TRACE", {"SPUMONI_CALL_TRACE": "1", "SPX_PHASE_TRACE": "1"})
    for e in (".lengths", ".pointers", ".report"):
        os.replace(f"{dm}/reads.fa{e}", f"{dm}/mapped{e}")
    run_ms("SPUMONI_MAP_OUTPUT=0 (plain writes, one writer thread per file)", {"SPUMONI_MAP_OUTPUT": "0"})
    for e in (".lengths", ".pointers", ".report"):
        same = subprocess.run(["cmp", f"{dm}/reads.fa{e}", f"{dm}/mapped{e}"]).returncode == 0
        print(f"   cmp {e} (files' tails as memory against plain writes): {'identical' if same else 'DIFFERENT'}", flush=True)
    import shutil
    shutil.rmtree(dm)

# ---- run -m: the reads digested on the device before the walk (the declared C3 pipeline: a minimizer-digested index) ----
if os.environ.get("E2E_DIGEST", "1") != "0":
    from spumoni_amd import capi
    dg = d + "/dig"; os.makedirs(dg, exist_ok=True)
    k, w = 4, 11
    print("   (digesting the genomes)", flush=True)
    dig = capi.digester(0)
    parts = []
    for g in genomes:
        for seq in (g, synth.revcomp(g)):
            dd, _ = dig.digest_host(capi.SPX_DIGEST_PROMOTED, k, w, seq, np.array([0, seq.size], dtype=np.uint64))
            parts.append(dd.copy())
    dig.close()
    print("   (indexing the digested text)", flush=True)
    raw_d = synth.index_from_text(torch.from_numpy(np.concatenate(parts)).cuda(), with_samples=False).cpu()
    open(f"{dg}/ref.bin", "w").write(">x\n")
    raw_d.write_raw_files(f"{dg}/ref.bin")
    write_null_db(f"{dg}/ref.bin.pmlnulldb", 3.0, [1, 2, 3, 3, 3, 3, 3])
    os.symlink(f"{d}/reads.fa", f"{dg}/reads.fa")

    def run_m(tag, extra_env):
        env = dict(os.environ, **extra_env)
        t0 = time.time()
        print(f"   (starting: {tag})", flush=True)
        r = subprocess.run(["timeout", "-s", "ABRT", os.environ.get("E2E_M_TIMEOUT", "120"), f"{ROOT}/spumoni_amd/bin/spumoni", "run", "-r", f"{dg}/ref", "-p", f"{dg}/reads.fa", "-P", "-c", "-m"], capture_output=True, env=env)
        if r.returncode != 0:
            print("   FAILED rc", r.returncode, r.stderr.decode(errors="replace")[-1500:], flush=True)
            return
        dt = time.time() - t0
        err = r.stderr.decode().replace("\033[32m", "").replace("\033[0m", "")
        assert r.returncode == 0, err
        import re
        secs = [float(x) for x in re.findall(r"done\.\s+\(([0-9.]+) sec\)", err)]
        load_s, proc_s = (secs + [0, 0])[:2]
        print(f"== -P -c -m (k=4, w=11), {nreads} reads, {tag}: {dt:.2f}s wall; loading the index {load_s:.3f}s, processing the patterns {proc_s:.3f}s = "
              f"{nreads / max(proc_s, 1e-9) / 1e6:.2f} M reads/s (pseudo_lengths {os.path.getsize(dg + '/reads.fa.pseudo_lengths') / 1e6:.0f} MB, "
              f"report {os.path.getsize(dg + '/reads.fa.report') / 1e6:.0f} MB)")
        for l in err.splitlines():
            if "[timing]" in l:
                print("   ", l.strip())
        sys.stdout.flush()

    run_m("SPUMONI_CACHE=write", {"SPUMONI_CACHE": "write"})
    run_m("flat-layout cache", {})
    run_m("flat-layout cache, again", {})
    run_m("SPUMONI_MAP_OUTPUT=0", {"SPUMONI_MAP_OUTPUT": "0"})
    import shutil
    shutil.rmtree(dg)

# ---- CPU: the oracle harness, file to file, one thread (the reference's -t 1 shape) ----
run("GPU CLI on the CPU sample", f"{d}/sample.fa", {})
for ext in (".pseudo_lengths", ".report"):
    os.replace(f"{d}/sample.fa{ext}", f"{d}/gpu_sample{ext}")
orc = os.path.join(ROOT, "oracle", "orc_run")
if os.path.exists(orc):
    t0 = time.time()
    r = subprocess.run([orc, f"{d}/ref.fa", f"{d}/sample.fa", "P", "0", "1", "150", "n"], capture_output=True)
    dt = time.time() - t0
    assert r.returncode == 0, r.stderr.decode()
    print(f"== CPU oracle harness (1 thread, index load included): {dt:.2f}s wall for {ncpu} reads = {ncpu/dt/1e3:.1f} k reads/s, "
          f"files written")
    for ext in (".pseudo_lengths", ".report"):
        same = subprocess.run(["cmp", f"{d}/sample.fa{ext}", f"{d}/gpu_sample{ext}"]).returncode == 0
        print(f"   cmp {ext}: {'identical' if same else 'DIFFERENT'}")
