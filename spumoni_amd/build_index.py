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
  .fa.pmlnulldb/.msnulldb  empirical null database (src/emp_null_database.cpp:19-110), with the KS-statistic
                           threshold of compute_ms_pml.cpp:1549-1663 in its second field
  spumoni_null_reads.fa    (next to the prefix) the null reads the statistics were taken from, chosen the way
                           src/refbuilder.cpp:83-127 / :234-270 chooses them (srand(0), rand() of glibc)
  .fa.doc / .fa.fdi        document array (src/doc_array.cpp) when --doc

The text is each sequence followed by its reverse complement, no separators, upper-cased (the order
src/refbuilder.cpp:100-180 writes them in); the BWT / thresholds / samples are computed here by suffix sorting, not by
PFP, and nothing here has been compared with an upstream-built index (none exists offline).
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


def _open_maybe_gz(path):
    """The reference reads its input FASTA files through gzopen (src/refbuilder.cpp:93): plain or gzip-compressed."""
    with open(path, "rb") as f:
        magic = f.read(2)
    if magic == b"\x1f\x8b":
        import gzip

        return gzip.open(path, "rb")
    return open(path, "rb")


def read_fasta(path, upper=True):
    seqs, cur = [], []
    with _open_maybe_gz(path) as f:
        for line in f:
            if line.startswith(b">"):
                if cur:
                    seqs.append(b"".join(cur))
                cur = []
            else:
                cur.append(line.strip())
    if cur:
        seqs.append(b"".join(cur))
    return [np.frombuffer(s.upper() if upper else s, dtype=np.uint8) for s in seqs if len(s)]


class GlibcRand:
    """rand() of glibc after srand(seed) (the TYPE_3 additive-feedback generator, r[i] = r[i-3] + r[i-31], 310 values
    discarded, top 31 bits returned): the reference picks its null reads and the null windows of its KS fit with it
    (src/refbuilder.cpp:83, 116; src/ks_test.cpp:113), so the same input files give the same reads here."""

    def __init__(self, seed=0):
        seed = seed & 0xFFFFFFFF
        if seed == 0:
            seed = 1  # srand(0) is srand(1)
        r = [seed]
        for i in range(1, 31):
            # 16807 * r mod (2^31 - 1), computed the way glibc does (signed 32-bit, Schrage)
            word = r[-1] if r[-1] < 2**31 else r[-1] - 2**32
            hi, lo = int(word / 127773), 0
            lo = word - hi * 127773  # C division truncates towards zero, % follows
            word = 16807 * lo - 2836 * hi
            if word < 0:
                word += 2147483647
            r.append(word & 0xFFFFFFFF)
        self._r = r + r[:3]  # r[31..33] = r[0..2]
        for _ in range(310):
            self._next()

    def _next(self):
        r = self._r
        v = (r[-31] + r[-3]) & 0xFFFFFFFF
        r.append(v)
        del r[0]
        return v

    def rand(self):
        return self._next() >> 1


NULL_READ_CHUNK, NUM_NULL_READS, NULL_READ_BOUND = 150, 800, 1000  # include/spumoni_main.hpp:65-67


def null_reads_from_list(files_seqs, rng):
    """The null reads of a file list (src/refbuilder.cpp:83-127): from every sequence (upper-cased), while fewer than
    1000 have been taken, 100 random 150-character substrings (25 once 800 are there); a sequence of at most 150
    characters is taken whole (and counted, whatever the count is)."""
    reads = []
    for seqs in files_seqs:
        for s in seqs:
            grab = 25 if len(reads) >= NUM_NULL_READS else 100
            go = len(reads) < NULL_READ_BOUND
            i = 0
            while i < grab and go and s.size > NULL_READ_CHUNK:
                at = rng.rand() % (s.size - NULL_READ_CHUNK)
                reads.append(s[at: at + NULL_READ_CHUNK])
                go = len(reads) < NULL_READ_BOUND
                i += 1
            if s.size <= NULL_READ_CHUNK:
                reads.append(s)
    return reads


def null_reads_from_fasta(seqs, rng):
    """The null reads of a single FASTA file (src/refbuilder.cpp:234-270): as above, but the sequences are read as
    they are in the file (no upper-casing), a substring that contains 'N' is drawn and dropped, and reading stops after
    the sequence in which a sampled substring brings the count to 1000 (whole short sequences do not stop it)."""
    reads = []
    go = True  # go_for_extraction (:247): only the sampling branch ever clears it -- sequences of at most 150 characters are
    #            counted without touching it, so a file of many short sequences yields more than 1000 reads
    for s in seqs:
        if not go:
            break
        grab = 25 if len(reads) >= NUM_NULL_READS else 100
        i = 0
        while i < grab and go and s.size > NULL_READ_CHUNK:
            at = rng.rand() % (s.size - NULL_READ_CHUNK)
            piece = s[at: at + NULL_READ_CHUNK]
            if not (piece == ord("N")).any():
                reads.append(piece)
                go = len(reads) < NULL_READ_BOUND
            i += 1
        if s.size <= NULL_READ_CHUNK:
            reads.append(s)
    return reads


def ks_statistic(pos_stats, null_stats):
    """KSTest::run_test (src/ks_test.cpp:80-104): the largest amount by which the null sample's empirical CDF lies above
    the positive sample's (one-sided on purpose: only a shift of the positives to the right counts), over the values
    0, 1, ... up to the first at which either CDF reaches 1."""
    pos_stats = np.asarray(pos_stats, dtype=np.int64)
    null_stats = np.asarray(null_stats, dtype=np.int64)
    top = int(max(pos_stats.max(), null_stats.max()))
    pos_cdf = np.cumsum(np.bincount(pos_stats, minlength=top + 1)) / (pos_stats.size + 0.0)
    null_cdf = np.cumsum(np.bincount(null_stats, minlength=top + 1)) / (null_stats.size + 0.0)
    done = np.nonzero((pos_cdf >= 1.0) | (null_cdf >= 1.0))[0]
    last = int(done[0]) if done.size else top
    return float(max(0.0, (null_cdf[: last + 1] - pos_cdf[: last + 1]).max()))


def run_kstest(lengths, null_stats, bin_size, rng):
    """KSTest::run_kstest (src/ks_test.cpp:106-134): the read's statistics in windows of bin_size (the last window
    takes what is left when less than two windows remain), each against a window of the null statistics that starts
    at rand() % (num_values - 2 * bin_size)."""
    lengths = np.asarray(lengths)
    nv = len(null_stats)
    if nv == 2 * bin_size:
        raise ValueError("the null database holds exactly two windows of statistics: the reference divides by zero here")
    out, start = [], 0
    while start < lengths.size:
        draw = rng.rand()
        null_pos = 0 if nv < 2 * bin_size else draw % (nv - 2 * bin_size)
        if lengths.size < bin_size:
            end = lengths.size
        else:
            end = start + bin_size if start + bin_size <= lengths.size - bin_size else lengths.size
        region = end - start
        # (the reference reads past the end of the null statistics when a window is longer than what is left of them:
        # undefined there, cut short here)
        out.append(ks_statistic(lengths[start:end], null_stats[null_pos: null_pos + region]))
        start += region
    return out


def ks_threshold(per_read_lengths, null_stats, bin_size, rng):
    """find_threshold_based_on_null_{pml,ms}_distribution (compute_ms_pml.cpp:1549-1663): the KS statistics of the
    null reads' own windows against the null database; threshold = mean + 3 standard deviations."""
    ks = []
    for lengths in per_read_lengths:
        if len(lengths):
            ks.extend(run_kstest(lengths, null_stats, bin_size, rng))
    total = 0.0
    for x in ks:
        total += x
    mean = total / len(ks)
    sq = 0.0
    for x in ks:
        sq += (x - mean) ** 2
    return mean + 3 * (sq / len(ks)) ** 0.5


def _int_vector(vals, width):
    vals = [int(v) & ((1 << width) - 1) for v in vals]
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
    """Bits per null statistic as the reference sizes them: max(ceil(log2(largest)), 1)
    (src/emp_null_database.cpp:41-44).  A largest value that is a power of two does not fit in that width and is stored
    cut to its low bits, there as here (the mean and the percentile are taken before the values are stored)."""
    mx = max([0] + [int(v) for v in vals])
    return max(int(np.ceil(np.log2(mx))), 1) if mx > 0 else 1


def percentile_value(stats):
    """largest value that occurs at least 5 times (src/emp_null_database.cpp:60-80)."""
    vals, counts = np.unique(np.asarray(stats), return_counts=True)
    ok = vals[counts >= 5]
    return float(ok.max()) if ok.size else 0.0


def write_null_db(path, stats, ks_stat_threshold=0.0):
    """EmpNullDatabase::serialize (src/emp_null_database.cpp:83-101): num_values, ks_stat_threshold, mean_null_stat,
    percentile_value, then the statistics as an sdsl int_vector."""
    stats = np.asarray(stats, dtype=np.uint64)
    with open(path, "wb") as f:
        f.write(struct.pack("<Qddd", stats.size, float(ks_stat_threshold), float(stats.mean()) if stats.size else 0.0,
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
    ap.add_argument("--null-reads", type=int, default=None, help="keep only the first N null reads (default: all, at most ~1000)")
    ap.add_argument("-w", "--window", type=int, default=150, help="bin size of the KS fit (50-400)")
    ap.add_argument("-m", "--minimizer-alphabet", action="store_true",
                    help="index the promoted-minimizer digestion of every sequence (files named <prefix>.bin*)")
    ap.add_argument("-a", "--dna-minimizer", action="store_true",
                    help="index the DNA-letter minimizer digestion of every sequence")
    ap.add_argument("--serialized", action="store_true",
                    help="also write <prefix>.thrbv.spumoni and <prefix>.thrbv.ms, the serialised indexes `spumoni run` loads "
                         "(compute_ms_pml.cpp:192-213, 517-542; stream layout unverified against upstream, Python loops: small inputs)")
    ap.add_argument("-K", "--small-window", type=int, default=4)
    ap.add_argument("-W", "--large-window", type=int, default=11)
    a = ap.parse_args(argv)
    if a.minimizer_alphabet and a.dna_minimizer:
        ap.error("only one of -m / -a")
    digest_kind = capi.SPX_DIGEST_PROMOTED if a.minimizer_alphabet else capi.SPX_DIGEST_DNA if a.dna_minimizer else 0
    digester = capi.digester(0) if digest_kind else None
    files, doc_ids = [a.ref] if a.ref else [], []
    if not a.ref and a.filelist:
        # "<path> [<document id>]" per line; the ids start at 1 and stay or grow by one (src/refbuilder.cpp:52-70)
        for ln in open(a.filelist):
            w = ln.split()
            if not w:
                continue
            files.append(w[0])
            if a.doc:
                if len(w) < 2:
                    doc_ids.append(len(files))  # no id column: one document per file
                elif not w[1].isdigit():
                    sys.exit(f"A document ID in the file_list is not an integer: {w[1]}")
                else:
                    doc_ids.append(int(w[1]))
        if a.doc and doc_ids:
            if doc_ids[0] != 1:
                sys.exit("The first ID in file_list must be 1")
            if any(b not in (x, x + 1) for x, b in zip(doc_ids, doc_ids[1:])):
                sys.exit("The IDs in the file_list must be staying constant or increasing by 1.")
    if not files:
        ap.error("give -r or -l")
    parts, doc_lengths, files_seqs = [], [], []
    for fi, fpath in enumerate(files):
        total = 0
        seqs = read_fasta(fpath)
        files_seqs.append(seqs)
        for s in seqs:
            pieces = [s] if a.no_rev_comp else [s, synth.revcomp(s)]
            for p in pieces:
                if digester is not None:  # every sequence is digested on its own, like a read
                    p, _ = digester.digest_host(digest_kind, a.small_window, a.large_window, p,
                                                np.array([0, p.size], dtype=np.uint64))
                    p = p.copy()
                parts.append(p)
                total += p.size
        if doc_ids and fi > 0 and doc_ids[fi] == doc_ids[fi - 1]:
            doc_lengths[-1] += total  # the same document as the file before
        else:
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
    if a.serialized:
        from spumoni_amd.sdsl_streams import write_thrbv

        #  these files carry upstream's names but not everything upstream's loader trusts: say so every time
        print("[build_index] WARNING: --serialized writes <prefix>.thrbv.spumoni / .thrbv.ms in a restatement of the sdsl-lite / "
              "r-index stream layout that is UNVERIFIED against an upstream-built file, and the rank / select support "
              "structures inside them are placeholders: this package's `spumoni run` rebuilds them and reads the files "
              "correctly; upstream `spumoni run` would deserialise them without complaint and answer wrongly.  Do not hand "
              "them to upstream.", file=sys.stderr)

        heads = np.maximum(raw.heads.numpy(), 1)
        write_thrbv(prefix + ".thrbv.spumoni", heads, raw.lens.numpy(), raw.thr.numpy())
        write_thrbv(prefix + ".thrbv.ms", heads, raw.lens.numpy(), raw.thr.numpy(), raw.ssa.numpy(), raw.esa.numpy())
    if a.doc:
        write_doc_array(prefix + ".doc", raw.doc_start.tolist(), raw.doc_end.tolist(), len(doc_lengths))
    # empirical null (compute_ms_pml.cpp:1409-1506): the null reads upper-cased, reversed, digested like the text;
    # their statistics are the database, and their own KS statistics against it give the threshold (:1549-1663).
    # MS first, then PML, one rand() sequence through both (src/spumoni.cpp:650-694).
    rng = GlibcRand(0)
    if a.ref:
        null_reads = null_reads_from_fasta(read_fasta(a.ref, upper=False), rng)
    else:
        null_reads = null_reads_from_list(files_seqs, rng)
    if a.null_reads is not None:
        null_reads = null_reads[: a.null_reads]
    null_path = os.path.join(os.path.dirname(os.path.abspath(prefix)), "spumoni_null_reads.fa")  # src/spumoni.cpp:586
    with open(null_path, "wb") as f:
        for i, rd in enumerate(null_reads):
            f.write(b">read_%d\n" % i + rd.tobytes() + b"\n")
    stats = {"ms": [0], "pml": [0]}
    ks_thr = {"ms": 0.0, "pml": 0.0}
    # (the statistics come from the library's path: asked of the library itself whether it has a device to run it on)
    if null_reads and capi.lib().spx_device_count() > 0:
        rev = [np.frombuffer(rd.tobytes().upper(), dtype=np.uint8)[::-1] for rd in null_reads]
        seqs = np.ascontiguousarray(np.concatenate(rev))
        offs = np.concatenate([[0], np.cumsum([r.size for r in rev])]).astype(np.uint64)
        if digester is not None:
            seqs, offs = digester.digest_host(digest_kind, a.small_window, a.large_window, seqs, offs)
            seqs, offs = seqs.copy(), offs.copy()
        ix = capi.Index.from_raw(raw, 0)
        for key, mode in (("ms", capi.SPX_MODE_MS), ("pml", capi.SPX_MODE_PML)):
            lengths = np.asarray(ix.query_host(mode, seqs, offs)["lengths"])
            if lengths.size:
                stats[key] = lengths
                per_read = [lengths[int(offs[q]): int(offs[q + 1])] for q in range(offs.size - 1)]
                # (the windows of the database are read back from its int_vector: src/ks_test.cpp:125)
                stored = lengths.astype(np.uint64) & np.uint64((1 << _width(lengths.tolist())) - 1)
                ks_thr[key] = ks_threshold(per_read, stored, a.window, rng)
    write_null_db(prefix + ".pmlnulldb", stats["pml"], ks_thr["pml"])
    write_null_db(prefix + ".msnulldb", stats["ms"], ks_thr["ms"])
    stats_pml, stats_ms = stats["pml"], stats["ms"]
    print(f"built {prefix}.*: n = {raw.n}, r = {raw.r}, documents = {len(doc_lengths)}, null reads = {len(null_reads)}, "
          f"null percentile PML = {percentile_value(stats_pml)} MS = {percentile_value(stats_ms)}, "
          f"KS threshold PML = {ks_thr['pml']:.4f} MS = {ks_thr['ms']:.4f}")


if __name__ == "__main__":
    main()
