"""bench.py with more than one rank, on ONE GPU: `--gpus N` makes bench.py its own launcher
(torch.distributed.run, one rank per GPU); BENCH_SHARE_GPUS=1 folds the ranks onto the device that
exists and BENCH_DIST_BACKEND=gloo carries the count reduction (RCCL refuses two ranks on one
device).  What is under test is everything but the transport: the respawn, RANK / LOCAL_RANK /
WORLD_SIZE handling, the sharding of the reads (weak: a batch per rank; strong: one read set cut
by shard.partition_reads), the count reduction and the line's per-rank figures
(reads are independent: /root/reference/src/compute_ms_pml.cpp:890-1024 carries no cross-read state)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--runs", "1500000", "--reads", "200000", "--steps", "2", "--no-cpu-baseline", "--no-extras"]


def _bench(args, share=False):
    env = dict(os.environ)
    env["SPX_INDEX_BUDGET_GB"] = "4"  # two ranks flatten on one device at the same time
    if share:
        env["BENCH_SHARE_GPUS"] = "1"
        env["BENCH_DIST_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + COMMON + args, env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]  # ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.timeout(1500)
def test_two_ranks_weak_counts_are_the_sum_of_the_single_runs():
    two = _bench(["--gpus", "2"], share=True)
    assert two["n_gpus"] == 2 and two["scaling"] == "weak"
    c2 = two["config"]["classified"]
    assert c2["reads"] == 2 * 200000
    singles = [_bench(["--gpus", "1", "--seed-offset", str(i)]) for i in range(2)]
    for key in ("reads", "bases", "FOUND", "NOT_PRESENT"):
        assert c2[key] == sum(s["config"]["classified"][key] for s in singles), key
    pr = two["per_rank_ms_per_step"]
    assert len(pr["ranks"]) == 2 and [r["reads"] for r in pr["ranks"]] == [200000, 200000]
    assert 0 < pr["min"] <= pr["max"]
    assert two["value"] > 0 and two["roofline"]["frac"] > 0
    # ... and the line carries the OTHER scaling too: one set of 200 000 reads cut in two, same protocol
    o = two["other_scaling"]
    assert o["scaling"] == "strong" and o["reads_per_step"] == 200000 and o["classified_reads"] == 200000
    assert sum(r["reads"] for r in o["per_rank_ms_per_step"]) == 200000 and o["value"] > 0


@pytest.mark.gpu
@pytest.mark.timeout(1500)
def test_two_ranks_strong_cut_one_read_set():
    one = _bench(["--gpus", "1", "--scaling", "strong"])
    two = _bench(["--gpus", "2", "--scaling", "strong"], share=True)
    assert two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert two["config"]["classified"] == one["config"]["classified"]  # the same 200 000 reads, cut in two
    assert two["config"]["classified"]["reads"] == 200000
    assert sum(r["reads"] for r in two["per_rank_ms_per_step"]["ranks"]) == 200000
    o = two["other_scaling"]  # a strong line carries the weak figure beside it
    assert o["scaling"] == "weak" and o["reads_per_step"] == 400000 and o["classified_reads"] == 400000
    assert "other_scaling" not in one  # (one rank: the two are the same job)
    # the weak line of one rank walks the same reads (seed 13): same counts again
    assert _bench(["--gpus", "1"])["config"]["classified"] == one["config"]["classified"]


@pytest.mark.gpu
def test_more_ranks_than_gpus_is_a_clear_error():
    import torch

    n = torch.cuda.device_count() + 1
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + COMMON, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    assert "wants GPU" in p.stderr and "the node has" in p.stderr


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_one_rank_under_torchrun_with_rccl():
    """The launcher the driver uses (python -m torch.distributed.run ... bench.py --gpus N) with the RCCL backend, at the
    one world size a one-GPU box allows: process-group initialisation on the device, the count all-reduce, the
    barriers and the all-gather of the per-rank step times go through RCCL (backend "nccl" IS RCCL on ROCm)."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SPX_INDEX_BUDGET_GB="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("BENCH_DIST_BACKEND", None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1"] + COMMON,
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=800)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["classified"]["reads"] == 200000
    assert d["config"]["classified"] == _bench(["--gpus", "1"])["config"]["classified"]
