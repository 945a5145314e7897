#!/usr/bin/env python3
"""Generates the committed golden fixtures (run from the repo root: python tests/golden/make_golden.py).

The reference ships no golden vectors and cannot be built offline, so these fixtures are
OURS: inputs made by spumoni_amd.synth, expected outputs computed by the CPU oracle
(oracle/), which is itself pinned by the brute-force KATs in tests/test_oracle_kat.py.
They freeze today's behaviour so that any later change of the oracle or of the HIP path
shows up as a diff against committed data, and they let the GPU tests check the HIP path
without re-deriving expectations on the box.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from spumoni_amd import synth  # noqa: E402
from tests import cases  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: (seed, text length, letters, extra read letters)
    "dna_single": (101, 1500, list(b"ACGT"), [ord("N")]),
    "dna_with_n": (102, 2500, list(b"ACGTN"), [ord("Z"), 0, 1, 2]),
    "promoted_alphabet": (103, 4000, [3, 4, 5, 60, 127, 128, 129, 200, 255], [2, 250]),
}


def main():
    for name, (seed, n, letters, extra) in CASES.items():
        raw, text = cases.real_case(seed, n, letters, ndocs=3)
        rng = np.random.default_rng(seed + 1000)
        seqs, offs = cases.reads_mixed(rng, text, letters, 60, 90, extra)
        orc = oracle.OracleIndex.from_raw(raw)
        pml, pdocs = orc.pml(seqs, offs, want_docs=True)
        ms = orc.ms(seqs, offs, want_docs=True, text=text)
        f, a, b, s = oracle.classify(pml, offs, 25, 5)
        st = orc.stats(seqs, offs)
        np.savez_compressed(
            os.path.join(OUT, name + ".npz"),
            heads=raw.heads.numpy(), lens=raw.lens.numpy(), thr=raw.thr.numpy(), ssa=raw.ssa.numpy(),
            esa=raw.esa.numpy(), doc_start=raw.doc_start.numpy(), doc_end=raw.doc_end.numpy(), text=text,
            seqs=seqs, offs=offs, pml=pml, pml_docs=pdocs, ms_pointers=ms["pointers"], ms_lengths=ms["lengths"],
            ms_docs=ms["docs"], cls_found=f, cls_above=a, cls_below=b, cls_sum=s, bin_width=25, max_value_thr=5,
            steps=st["steps"], jumps=st["jumps"], pred_jumps=st["pred_jumps"],
        )
        print(name, "r =", raw.r, "n =", raw.n, "reads =", offs.size - 1, "chars =", int(offs[-1]))
    # a statistical index (no text): zipf alphabet with bytes >= 128
    raw = synth.statistical_rlbwt(3000, 60, 3.0, seed=7, zipf=1.0, letters=list(range(100, 160)), with_samples=True, n_docs=5)
    seqs, offs = synth.simulate_reads(raw, 80, 50, seed=8, f_mis=0.1)
    seqs, offs = seqs.numpy(), offs.numpy()
    orc = oracle.OracleIndex.from_raw(raw)
    pml, pdocs = orc.pml(seqs, offs, want_docs=True)
    ms = orc.ms(seqs, offs, want_docs=True)
    np.savez_compressed(
        os.path.join(OUT, "statistical_zipf.npz"),
        heads=raw.heads.numpy(), lens=raw.lens.numpy(), thr=raw.thr.numpy(), ssa=raw.ssa.numpy(), esa=raw.esa.numpy(),
        doc_start=raw.doc_start.numpy(), doc_end=raw.doc_end.numpy(), seqs=seqs, offs=offs, pml=pml, pml_docs=pdocs,
        ms_pointers=ms["pointers"], ms_docs=ms["docs"],
    )
    print("statistical_zipf r =", raw.r)


if __name__ == "__main__":
    main()
