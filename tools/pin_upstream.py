#!/usr/bin/env python3
"""pin_upstream.py -- pins the oracle (and, with --gpu, the HIP path) to an UPSTREAM-built spumoni run.

Nothing in this repository has ever been compared with output of the real oma219/spumoni binary: it
cannot be built offline (sdsl-lite, r-index, ShapedSlp, bonsai are fetched by its CMake) and it ships
no test vectors, so every parity claim is "HIP == oracle" with the oracle pinned by mathematics only
(DESIGN.md 3).  This script is what closes that gap the day somebody can run upstream once.

What an external run must supply -- one directory <dir> holding

  index, built with `spumoni build -r <ref> -M -P [-d] -k [-m|-a|-n]` (-k keeps the raw files):
      ref.fa  (or ref.bin for -m)                      the file name `run -r <dir>/ref` resolves
      ref.fa.thrbv.spumoni, ref.fa.thrbv.ms            serialised indexes (PML / MS)
      ref.fa.bwt.heads  .bwt.len  .thr_pos  .ssa  .esa  raw run files (5-byte LE records)
      ref.fa.doc                                        document array          (only with -d)
      ref.fa.pmlnulldb  ref.fa.msnulldb                 null databases
      ref.fa.rawtext                                    OPTIONAL: the concatenated text PFP indexed,
                                                        one byte per character, no terminator (the CPU
                                                        oracle harness needs it for MS; the HIP CLI
                                                        rebuilds the text from the MS index without it)
  queries and upstream results, from `spumoni run -t 1 -r <dir>/ref -p <dir>/reads.fa <flags>`:
      reads.fa                                          the patterns
      expected/P/reads.fa.pseudo_lengths [.doc_numbers] [.report]        flags -P [-d] [-c]
      expected/M/reads.fa.lengths  .pointers [.doc_numbers] [.report]    flags -M [-d] [-c]
      run_flags.txt                                     one line: the digestion flag used: -n, -m or -a
                                                        (optionally "-K <k> -W <w>")

Checks (each prints PASS / FAIL / SKIP; exit code 1 on any FAIL):
  1. serialised-index reader == raw ingest: `spumoni dump-index` of .thrbv.spumoni / .thrbv.ms against the
     raw run files (heads with 0 -> 1, lengths, thresholds with thr_bv's zero-skipping, samples)
  2. oracle harness (oracle/orc_run, CPU) on reads.fa: every output file byte-identical to expected/
     -- this pins the oracle's _query restatement, the writers' byte format, BatchLoader, the
     classifier and (with -m / -a) the digestion assumptions B1-B5
  3. --gpu: the HIP-backed `spumoni run` on the same input, same comparison (needs an MI355X)
"""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_BIN = os.path.join(ROOT, "spumoni_amd", "bin", "spumoni")
ORC_RUN = os.path.join(ROOT, "oracle", "orc_run")

results = []


def report(name, status, detail=""):
    results.append((name, status))
    print(f"[{status:4s}] {name}" + (f": {detail}" if detail else ""), flush=True)


def five(path, stride=1, pick=0):
    raw = np.fromfile(path, dtype=np.uint8)
    rec = raw.reshape(-1, 5 * stride)[:, 5 * pick : 5 * pick + 5]
    out = np.zeros((rec.shape[0], 8), dtype=np.uint8)
    out[:, :5] = rec
    return out.view("<u8").reshape(-1)


def check_reader(prefix, mode):
    ser = prefix + (".thrbv.ms" if mode == "M" else ".thrbv.spumoni")
    name = f"1. serialised reader == raw ingest ({os.path.basename(ser)})"
    if not (os.path.exists(ser) and os.path.exists(prefix + ".bwt.heads")):
        return report(name, "SKIP", "serialised index or raw files missing")
    r = subprocess.run([HOST_BIN, "dump-index", ser, mode], capture_output=True)
    if r.returncode != 0:
        return report(name, "FAIL", "reader refused the file: " + r.stderr.decode().strip()[:300])
    got = {}
    for line in r.stdout.decode().splitlines():
        k, *v = line.split()
        got[k] = np.array(v, dtype=np.uint64) if k != "n" else v
    heads = np.fromfile(prefix + ".bwt.heads", dtype=np.uint8).astype(np.uint64)
    heads[heads <= 1] = 1  # ms_rle_string.hpp:249-253
    lens = five(prefix + ".bwt.len")
    thr = five(prefix + ".thr_pos")
    bad = []
    if not np.array_equal(got.get("heads"), heads):
        bad.append("heads")
    if not np.array_equal(got.get("lens"), lens):
        bad.append("lens")
    # thr_bv drops zero thresholds and maps the k-th c-run to the (k-1)-th stored one (thresholds_ds.hpp:
    # 421-423, 484-488): the reader hands back values that flatten to the same THR[] -- compare those
    def effective(h, t):
        out = np.zeros_like(t)
        for c in np.unique(h):
            idx = np.flatnonzero(h == c)
            stored = t[idx][t[idx] != 0]
            k = min(len(idx) - 1, len(stored))
            out[idx[1 : 1 + k]] = stored[:k]
        return out
    if "thr" not in got or not np.array_equal(effective(heads, got["thr"]), effective(heads, thr)):
        bad.append("thr")
    if mode == "M":
        n = int(lens.sum())
        for ext, key in ((".ssa", "ssa"), (".esa", "esa")):
            right = five(prefix + ext, 2, 1)
            want = np.where(right > 0, right - 1, n - 1).astype(np.uint64)  # compute_ms_pml.cpp:433
            if not np.array_equal(got.get(key), want):
                bad.append(key)
    report(name, "FAIL" if bad else "PASS", "differs in " + ", ".join(bad) if bad else f"r = {heads.size}")


def run_and_compare(tag, binary_kind, d, prefix, mode, digest_flag, kw, text):
    exp = os.path.join(d, "expected", mode)
    name = f"{tag} mode -{mode} {digest_flag}"
    if not os.path.isdir(exp):
        return report(name, "SKIP", f"{exp} missing")
    files = sorted(os.listdir(exp))
    use_doc = any(f.endswith(".doc_numbers") for f in files)
    rep = any(f.endswith(".report") for f in files)
    if mode == "M" and text is None and binary_kind == "oracle":
        files = [f for f in files if not f.endswith(".lengths") and not f.endswith(".report")]
    work = tempfile.mkdtemp(prefix="pin_")
    try:
        reads = os.path.join(work, "reads.fa")
        shutil.copy(os.path.join(d, "reads.fa"), reads)
        if binary_kind == "oracle":
            cmd = [ORC_RUN, prefix, reads, mode, "1" if use_doc else "0", "1" if rep else "0", "150", digest_flag[1]]
            if mode == "M":
                if text is None:
                    return report(name, "SKIP", "MS in the oracle harness needs ref.fa.rawtext")
                cmd.append(text)
            cmd += kw
            env = os.environ
        else:
            ref = prefix[: -len(".bin")] if prefix.endswith(".bin") else prefix[: -len(".fa")]
            cmd = [HOST_BIN, "run", "-r", ref, "-p", reads, "-" + mode, digest_flag] + (["-d"] if use_doc else []) + \
                  (["-c"] if rep else []) + [x.replace("--k", "-K").replace("--w", "-W") for x in kw]
            env = dict(os.environ, SPUMONI_CACHE="off")
            if text:
                env["SPUMONI_TEXT"] = text
        r = subprocess.run(cmd, capture_output=True, env=env)
        if r.returncode != 0:
            return report(name, "FAIL", "exit %d: %s" % (r.returncode, r.stderr.decode().strip()[-300:]))
        bad = []
        for f in files:
            mine = os.path.join(work, f)
            if not os.path.exists(mine):
                bad.append(f + " (not written)")
            elif open(mine, "rb").read() != open(os.path.join(exp, f), "rb").read():
                bad.append(f)
        report(name, "FAIL" if bad else "PASS", ("differs: " + ", ".join(bad)) if bad else f"{len(files)} files identical")
    finally:
        shutil.rmtree(work, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("dir")
    ap.add_argument("--gpu", action="store_true", help="also run the HIP-backed CLI (needs an MI355X)")
    a = ap.parse_args()
    d = os.path.abspath(a.dir)
    flags = (open(os.path.join(d, "run_flags.txt")).read().split() if os.path.exists(os.path.join(d, "run_flags.txt")) else ["-n"])
    digest_flag = next((f for f in flags if f in ("-n", "-m", "-a")), "-n")
    kw = []
    for opt, name in (("-K", "--k"), ("-W", "--w")):
        if opt in flags:
            kw += [name, flags[flags.index(opt) + 1]]
    prefix = os.path.join(d, "ref.bin" if digest_flag == "-m" else "ref.fa")
    text = prefix + ".rawtext" if os.path.exists(prefix + ".rawtext") else None
    for mode in ("P", "M"):
        check_reader(prefix, mode)
    for mode in ("P", "M"):
        run_and_compare("2. oracle harness vs upstream", "oracle", d, prefix, mode, digest_flag, kw, text)
    if a.gpu:
        for mode in ("P", "M"):
            run_and_compare("3. HIP CLI vs upstream", "gpu", d, prefix, mode, digest_flag, kw, text)
    nfail = sum(1 for _, s in results if s == "FAIL")
    npass = sum(1 for _, s in results if s == "PASS")
    print(f"{npass} passed, {nfail} failed, {len(results) - npass - nfail} skipped")
    if npass and not nfail:
        print("every supplied artefact agrees: record this run (upstream commit, command lines, this output) in "
              "DESIGN.md 3 and drop 'parity unpinned' for the rows it covers")
    sys.exit(1 if nfail else 0)


if __name__ == "__main__":
    main()
