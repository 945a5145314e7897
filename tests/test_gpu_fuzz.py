"""-m gpu: many small random indexes x random layout knobs, all four query variants against the
oracle.  Cheap insurance for the paths that depend on the index's shape: compact / general rows,
16-byte fat digests and their escape, every fat block size (uniform and per letter), long runs, rare letters, bytes >= 128,
consistent and inconsistent thresholds."""
import numpy as np
import pytest
import torch

from spumoni_amd import capi, synth
from tests.test_gpu_parity import _compare_all

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    capi.lib()
    return 0


# SPX_FUZZ_SEEDS seeds from SPX_FUZZ_FIRST on (default: 0..59; a longer sweep is a matter of two environment variables)
_FIRST = int(__import__("os").environ.get("SPX_FUZZ_FIRST", "0"))


@pytest.mark.parametrize("seed", range(_FIRST, _FIRST + int(__import__("os").environ.get("SPX_FUZZ_SEEDS", "60"))))
def test_random_index_shapes(gpu, oracle_mod, seed, monkeypatch):
    rng = np.random.default_rng(1000 + seed)
    sigma = int(rng.choice([3, 4, 5, 17, 60, 200]))
    lo = int(rng.choice([2, 3, 60, 120]))  # some alphabets reach past 127 (signed-char quirk)
    letters = sorted(rng.choice(np.arange(lo, 256), size=min(sigma, 256 - lo), replace=False).tolist())
    r = int(rng.choice([50, 400, 3000, 20000]))
    mean_run = float(rng.choice([1.0, 2.0, 8.0, 40.0]))
    raw = synth.statistical_rlbwt(r, len(letters), mean_run, seed=seed, letters=letters, zipf=float(rng.choice([0.0, 1.0])),
                                  with_samples=True, n_docs=int(rng.choice([1, 3, 200])))
    if seed % 4 == 1:  # a few very long runs: general row encoding, offsets that escape the fat digest
        lens = raw.lens.clone()
        pick = torch.from_numpy(rng.choice(np.arange(1, raw.r), size=max(1, raw.r // 50), replace=False))
        lens[pick] = torch.from_numpy(rng.integers(1 << 16, 1 << 18, size=pick.numel()))
        raw = synth.raw_from_runs(raw.heads, lens, seed, with_samples=True, n_docs=3)
    if seed % 4 == 2:  # thresholds anywhere: every step still defined upstream (Appendix C1 general path)
        nz = raw.thr > 0
        raw.thr = torch.where(nz, torch.from_numpy(rng.integers(1, raw.n + 1, size=raw.r)), raw.thr)
    knob = int(rng.integers(0, 3))
    if knob == 0:  # one block size for every letter
        monkeypatch.setenv("SPX_FAT_BSHIFT", str(int(rng.integers(0, 8))))
    elif knob == 1:  # per-letter block sizes: table density and the exponent of the letters' run shares
        monkeypatch.setenv("SPX_FAT_SLOTS_PER_RUN", str(float(rng.choice([0.02, 0.3, 1.5, 6.0, 40.0]))))
        monkeypatch.setenv("SPX_FAT_ALPHA", str(float(rng.choice([0.0, 0.5, 0.7, 1.0]))))
    if rng.random() < 0.25:
        monkeypatch.setenv("SPX_FAT_ALL_ESC", "1")
    if rng.random() < 0.25:
        monkeypatch.setenv("SPX_ROWS_WIDE", "1")
    nreads, m = int(rng.choice([1, 70, 900])), int(rng.choice([1, 9, 64, 300]))
    seqs, offs = synth.simulate_reads(raw, nreads, m, seed=seed, positive_fraction=float(rng.choice([0.0, 0.5, 1.0])))
    seqs = seqs.cpu().numpy().copy()
    absent = [c for c in range(2, 256) if c not in letters][:3]
    if absent and seed % 3 == 0:
        seqs[rng.random(seqs.size) < 0.03] = absent[0]  # letters that do not occur in the index
    _compare_all(oracle_mod, raw, None, seqs, offs.cpu().numpy())
