"""Committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py).
CPU: the oracle still reproduces them.  GPU (-m gpu): the HIP path reproduces them."""
import glob
import os

import numpy as np
import pytest
import torch

from spumoni_amd import synth

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz")))


def _raw(g):
    t = lambda k: torch.from_numpy(g[k]) if k in g else None  # noqa: E731
    return synth.RawIndex(heads=t("heads"), lens=t("lens"), thr=t("thr"), n=int(g["lens"].sum()), ssa=t("ssa"),
                          esa=t("esa"), doc_start=t("doc_start"), doc_end=t("doc_end"), text=t("text"))


def test_fixtures_exist():
    assert len(GOLD) >= 4


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_oracle_reproduces_golden(oracle_mod, path):
    g = np.load(path)
    orc = oracle_mod.OracleIndex.from_raw(_raw(g))
    pml, docs = orc.pml(g["seqs"], g["offs"], want_docs=True)
    assert np.array_equal(pml, g["pml"]) and np.array_equal(docs, g["pml_docs"])
    text = g["text"] if "text" in g else None
    ms = orc.ms(g["seqs"], g["offs"], want_docs=True, text=text)
    assert np.array_equal(ms["pointers"], g["ms_pointers"]) and np.array_equal(ms["docs"], g["ms_docs"])
    if text is not None:
        assert np.array_equal(ms["lengths"], g["ms_lengths"])
        f, a, b, s = oracle_mod.classify(pml, g["offs"], int(g["bin_width"]), int(g["max_value_thr"]))
        assert np.array_equal(f, g["cls_found"]) and np.array_equal(a, g["cls_above"])
        assert np.array_equal(b, g["cls_below"]) and np.array_equal(s, g["cls_sum"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_hip_reproduces_golden(path):
    from spumoni_amd import capi

    g = np.load(path)
    ix = capi.Index.from_raw(_raw(g), 0)
    has_text = "text" in g
    cl = (int(g["bin_width"]), int(g["max_value_thr"])) if has_text else None
    got = ix.query_host(capi.SPX_MODE_PML, g["seqs"], g["offs"], want_docs=True, classify=cl)
    assert np.array_equal(got["lengths"], g["pml"]) and np.array_equal(got["docs"], g["pml_docs"])
    if has_text:
        assert np.array_equal(got["class"]["above"], g["cls_above"])
        assert np.array_equal(got["class"]["below"], g["cls_below"])
        assert np.array_equal(got["class"]["sum_max"], g["cls_sum"])
        st = ix.last_stats()
        assert (st["steps"], st["jumps"], st["pred_jumps"]) == (int(g["steps"]), int(g["jumps"]), int(g["pred_jumps"]))
    got = ix.query_host(capi.SPX_MODE_MS, g["seqs"], g["offs"], want_lengths=has_text, want_docs=True)
    assert np.array_equal(got["pointers"], g["ms_pointers"]) and np.array_equal(got["docs"], g["ms_docs"])
    if has_text:
        assert np.array_equal(got["lengths"], g["ms_lengths"])
