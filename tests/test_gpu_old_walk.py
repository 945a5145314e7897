"""-m gpu: the state-machine walk (k_walk_lanes) on the inputs of the plain-walk tests.

Reads over compact rows are walked by k_walk_fast; k_walk_lanes stays for the general row encoding, the chunked
passes and -- with the one-bit-per-character length output and k_expand_lengths -- the reads that fall back from
a chunked walk.  SPX_OLD_WALK=1 (read once per process) routes every plain walk through it, so a subset of
the parity / fuzz / golden tests is run again in a process that has it set: both bodies are held against the
oracle on the same inputs.  (test_long_runs_and_far_thresholds is left out here: its index has the general row encoding,
which k_walk_lanes walks in the ordinary run already.)"""
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
                        "-k", "not scale and not long_runs_and_far_thresholds"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=1700)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert " passed" in p.stdout
