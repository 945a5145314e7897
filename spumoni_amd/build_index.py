"""Self-contained index builder for small / medium references (tooling).

`spumoni build` of the reference shells out to PFP / pfp-thresholds / RePair binaries that
are not available offline and is out of scope of this package.  This tool
exists so that the run path can be exercised end to end from a FASTA file without them:

    python -m spumoni_amd.build_index -r genome.fa -o out/prefix            # one document
    python -m spumoni_amd.build_index -l filelist.txt -o out/prefix --doc   # one document per file

It writes exactly the files `spumoni run` (ours) consumes, under `<prefix>.fa*`:
  .fa                      the concatenated text as FASTA (run's validate() wants it to exist)
  .fa.bwt.heads/.bwt.len   run-length BWT                (formats: SURVEY Appendix A.1)
  .fa.thr_pos              thresholds (first arg-min LCP between same-letter runs)
  .fa.ssa / .fa.esa        suffix-array samples at run starts / ends
  .fa.rawtext              the indexed text (MS length extension; replaces <ref>.slp)
  .fa.pmlnulldb/.msnulldb  empirical null database (src/emp_null_database.cpp:19-110)
  .fa.doc / .fa.fdi        document array (src/doc_array.cpp) when --doc

The text convention is OURS (each sequence followed by its reverse complement, no
separators, upper-cased): indexes built here are not byte-identical to upstream-built ones.
The suffix array is built by prefix doubling on the GPU when one is present (torch), else on
the CPU; the null statistics are computed with the HIP path itself.
"""
from __future__ import annotations

import argparse
import os
import struct
import sys

import numpy as np
import torch

from . import capi, synth


def read_fasta(path):
    seqs, cur = [], []
    with open(path, "rb") as f:
        for line in f:
            if line.startswith(b">"):
                if cur:
                    seqs.append(b"".join(cur))
                cur = []
            else:
                cur.append(line.strip())
    if cur:
        seqs.append(b"".join(cur))
    return [np.frombuffer(s.upper(), dtype=np.uint8) for s in seqs if len(s)]


def _int_vector(vals, width):
    vals = [int(v) for v in vals]
    bits = len(vals) * width
    words = [0] * ((bits + 63) // 64)
    for i, v in enumerate(vals):
        bit = i * width
        wi, sh = bit >> 6, bit & 63
        words[wi] |= (v << sh) & 0xFFFFFFFFFFFFFFFF
        if sh + width > 64:
            words[wi + 1] |= v >> (64 - sh)
    return struct.pack("<QB", bits, width) + b"".join(struct.pack("<Q", w) for w in words)


def _width(vals):
    mx = max([1] + [int(v) for v in vals])
    w = 1
    while (1 << w) <= mx:
        w += 1
    return w


def percentile_value(stats):
    """largest value that occurs at least 5 times (src/emp_null_database.cpp:60-80)."""
    vals, counts = np.unique(np.asarray(stats), return_counts=True)
    ok = vals[counts >= 5]
    return float(ok.max()) if ok.size else 0.0


def write_null_db(path, stats):
    stats = np.asarray(stats, dtype=np.uint64)
    with open(path, "wb") as f:
        f.write(struct.pack("<Qddd", stats.size, 0.0, float(stats.mean()) if stats.size else 0.0,
                            percentile_value(stats)))
        f.write(_int_vector(stats.tolist(), _width(stats.tolist())))


def write_doc_array(path, doc_start, doc_end, ndocs):
    w = max(1, int(np.ceil(np.log2(max(ndocs, 2)))))
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(doc_start)))
        f.write(_int_vector(doc_start, w))
        f.write(_int_vector(doc_end, w))


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m spumoni_amd.build_index")
    ap.add_argument("-r", "--ref", help="single FASTA file")
    ap.add_argument("-l", "--filelist", help="text file with one FASTA path per line (one document each)")
    ap.add_argument("-o", "--output", required=True, help="output prefix")
    ap.add_argument("--no-rev-comp", action="store_true")
    ap.add_argument("--doc", action="store_true", help="also write the document array")
    ap.add_argument("--null-reads", type=int, default=800)
    ap.add_argument("-m", "--minimizer-alphabet", action="store_true",
                    help="index the promoted-minimizer digestion of every sequence (files named <prefix>.bin*)")
    ap.add_argument("-a", "--dna-minimizer", action="store_true",
                    help="index the DNA-letter minimizer digestion of every sequence")
    ap.add_argument("-K", "--small-window", type=int, default=4)
    ap.add_argument("-W", "--large-window", type=int, default=11)
    a = ap.parse_args(argv)
    if a.minimizer_alphabet and a.dna_minimizer:
        ap.error("only one of -m / -a")
    digest_kind = capi.SPX_DIGEST_PROMOTED if a.minimizer_alphabet else capi.SPX_DIGEST_DNA if a.dna_minimizer else 0
    digester = capi.digester(0) if digest_kind else None
    files = [a.ref] if a.ref else [ln.split()[0] for ln in open(a.filelist) if ln.strip()]
    if not files:
        ap.error("give -r or -l")
    parts, doc_lengths = [], []
    for fpath in files:
        total = 0
        for s in read_fasta(fpath):
            pieces = [s] if a.no_rev_comp else [s, synth.revcomp(s)]
            for p in pieces:
                if digester is not None:  # every sequence is digested on its own, like a read
                    p, _ = digester.digest_host(digest_kind, a.small_window, a.large_window, p,
                                                np.array([0, p.size], dtype=np.uint64))
                    p = p.copy()
                parts.append(p)
                total += p.size
        doc_lengths.append(total)
    text = np.concatenate(parts)
    if text.min() < 2:
        sys.exit("the text contains bytes 0/1, which are reserved for the terminator")
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    raw = synth.index_from_text(torch.from_numpy(text).to(dev), doc_lengths=doc_lengths).cpu()
    prefix = a.output + (".bin" if a.minimizer_alphabet else ".fa")  # src/spumoni.cpp:744-747
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    with open(prefix, "wb") as f:
        f.write(b">concatenated\n" + text.tobytes() + b"\n")
    raw.write_raw_files(prefix)
    text.tofile(prefix + ".rawtext")
    with open(prefix + ".fdi", "w") as f:
        for i, ln in enumerate(doc_lengths):
            f.write(f"group_{i + 1}\t{ln}\n")
    if a.doc:
        write_doc_array(prefix + ".doc", raw.doc_start.tolist(), raw.doc_end.tolist(), len(doc_lengths))
    # empirical null: 150-bp substrings, reversed (compute_ms_pml.cpp:1424-1426, 1463-1465)
    rng = np.random.default_rng(0)
    L = 150
    stats_pml, stats_ms = [0], [0]
    if text.size > L and torch.cuda.is_available():
        starts = rng.integers(0, text.size - L, size=a.null_reads)
        reads = text[starts[:, None] + np.arange(L)[None, :]][:, ::-1]
        seqs = np.ascontiguousarray(reads.reshape(-1))
        offs = np.arange(a.null_reads + 1, dtype=np.uint64) * L
        ix = capi.Index.from_raw(raw, 0)
        stats_pml = ix.query_host(capi.SPX_MODE_PML, seqs, offs)["lengths"]
        stats_ms = ix.query_host(capi.SPX_MODE_MS, seqs, offs)["lengths"]
    write_null_db(prefix + ".pmlnulldb", stats_pml)
    write_null_db(prefix + ".msnulldb", stats_ms)
    print(f"built {prefix}.*: n = {raw.n}, r = {raw.r}, documents = {len(doc_lengths)}, "
          f"null percentile PML = {percentile_value(stats_pml)} MS = {percentile_value(stats_ms)}")


if __name__ == "__main__":
    main()
