"""-m gpu: the state-machine walk (k_walk_lanes) on the inputs of the plain-walk tests.

Reads over compact rows are walked by k_walk_fast; k_walk_lanes stays for the general row encoding, the chunked
passes and -- with the one-bit-per-character length output and k_expand_lengths -- the reads that fall back from
a chunked walk.  SPX_OLD_WALK=1 (read once per process) routes every plain walk through it, so a subset of
the parity / fuzz / golden tests is run again in a process that has it set: both bodies are held against the
oracle on the same inputs -- test_long_runs_and_far_thresholds included (its long runs are laid out as pieces since
round 3, so the ordinary run walks it with k_walk_fast as well; it was left out while it took 230 s, which was its own
numpy loop: profiles/r03_slow_test_probe.txt)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.timeout(1800)
def test_parity_suite_on_the_state_machine_walk():
    env = dict(os.environ)
    env["SPX_OLD_WALK"] = "1"
    env["SPX_FUZZ_SEEDS"] = "16"  # (the first sixteen shapes cover every knob; the full sixty run on the shipped kernel)
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        "tests/test_gpu_parity.py", "tests/test_golden.py",
                        "tests/test_gpu_fuzz.py",
                        "-k", "not scale"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=1700)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert " passed" in p.stdout


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_long_matching_read_expands_in_linear_time():
    """k_expand_lengths rescanned the bit mask from every group of 8 inside a reset-free stretch -- a read
    that matches for g characters cost g^2 / 512 loads on at most 64 lanes (seconds at Mbp scale).  The first reset
    after a position is now remembered across a lane's groups.  Reads that are one exact match of 0.2 and 1.6 Mbp, on
    the state-machine walk (the only one that writes the mask): right, and the time grows with the length, not its square."""
    code = r'''
import time, numpy as np, torch
from spumoni_amd import capi, synth
import oracle
rng = np.random.default_rng(1)
text = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=2_000_000)
raw = synth.index_from_text(torch.from_numpy(text.copy()).cuda(), with_samples=False).cpu()
orc = oracle.OracleIndex.from_raw(raw)
ix = capi.Index.from_raw(raw, 0)
ix.set_option("chunk_mode", 1)
secs = {}
for g in (200_000, 1_600_000):  # one lane walks the read (a microsecond per character); the expansion must not add a g^2 term
    reads = [text[1000:1000 + g], text[5:4000]]
    offs = np.concatenate([[0], np.cumsum([r.size for r in reads])]).astype(np.int64)
    seqs = np.concatenate(reads)
    d_seqs = capi.pad_seqs(torch.from_numpy(seqs).cuda()); d_offs = torch.from_numpy(offs).cuda()
    d_len = torch.empty(seqs.size + 8, dtype=torch.int32, device="cuda")
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        ix.query_device(capi.SPX_MODE_PML, d_seqs, d_offs, seqs.size, d_lengths=d_len)
        torch.cuda.synchronize(); secs[g] = time.time() - t0
    want = orc.pml(seqs, offs)
    assert np.array_equal(d_len[: seqs.size].cpu().numpy().view(np.uint32), want)
    assert want.max() >= g - 1000
print("SECONDS", secs)
assert secs[1_600_000] < 12 * secs[200_000], secs   # 8 x the characters: linear is 8 x, the old expansion was 64 x
'''
    env = dict(os.environ)
    env["SPX_OLD_WALK"] = "1"
    p = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=550)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
