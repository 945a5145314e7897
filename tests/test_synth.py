"""CPU checks of the synthetic generators against the oracle."""
import numpy as np
import torch

from spumoni_amd import synth


def test_statistical_rlbwt_is_a_valid_index(oracle_mod):
    raw = synth.statistical_rlbwt(5000, 253, 4.0, seed=5, zipf=1.0, with_samples=True, n_docs=4)
    assert raw.r == 5000 and int(raw.lens.sum()) == raw.n
    h = raw.heads.numpy()
    assert (h[1:] != h[:-1]).all()
    # thresholds lie in (end of previous same-letter run, start of this run]
    starts = (torch.cumsum(raw.lens, 0) - raw.lens).numpy()
    ends = starts + raw.lens.numpy() - 1
    last = {}
    for k in range(raw.r):
        c = max(int(h[k]), 1)
        t = int(raw.thr[k])
        if c in last:
            assert last[c] < t <= starts[k]
        else:
            assert t == 0
        last[c] = ends[k]
    orc = oracle_mod.OracleIndex.from_raw(raw)
    seqs, offs = synth.simulate_reads(raw, 400, 50, seed=1, positive_fraction=1.0, f_mis=0.0)
    st = orc.stats(seqs.numpy(), offs.numpy())
    # positive reads follow the walk: besides the first character of each read only bytes
    # >= 128 (signed-char quirk, ~11 % of a Zipf(1) alphabet of 253) take the jump branch
    hi = int((seqs.numpy() >= 128).sum())
    assert st["jumps"] <= 400 + hi
    assert st["pred_jumps"] <= 400
    seqs, offs = synth.simulate_reads(raw, 400, 50, seed=1, positive_fraction=0.0)
    st = orc.stats(seqs.numpy(), offs.numpy())
    assert st["jumps"] > 0.9 * st["steps"]


def test_dna_statistical_index_and_positive_reads(oracle_mod):
    raw = synth.statistical_rlbwt(3000, 4, 10.0, seed=6, letters=b"ACGT")
    orc = oracle_mod.OracleIndex.from_raw(raw)
    seqs, offs = synth.simulate_reads(raw, 200, 80, seed=2, positive_fraction=1.0, f_mis=0.05)
    st = orc.stats(seqs.numpy(), offs.numpy())
    assert 0.02 < st["jumps"] / st["steps"] < 0.12
    pml = orc.pml(seqs.numpy(), offs.numpy())
    assert pml.max() > 15
