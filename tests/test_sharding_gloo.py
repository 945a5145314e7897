"""The N>1 path on CPU: world_size-2 gloo.  The "device" behind each rank is the CPU
oracle here (tests may use it); what is under test is the sharding + count reduction that
bench.py / the multi-GPU harness run with RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from spumoni_amd import shard, synth
from tests import cases


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_partition_properties():
    rng = np.random.default_rng(0)
    for nreads in (0, 1, 2, 7, 100):
        lens = rng.integers(0, 50, size=nreads)
        offs = np.concatenate([[0], np.cumsum(lens)])
        for world in (1, 2, 3, 8):
            parts = shard.partition_reads(offs, world)
            assert len(parts) == world
            assert parts[0][0] == 0 and parts[-1][1] == nreads
            for (a, b), (c, d) in zip(parts, parts[1:]):
                assert a <= b == c <= d
            if nreads >= 50:
                chars = [offs[b] - offs[a] for a, b in parts]
                assert max(chars) - min(chars) <= 2 * 50


def _worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle

    raw, text = cases.real_case(5, 4000, list(b"ACGT"))
    rng = np.random.default_rng(9)
    seqs, offs = cases.reads_mixed(rng, text, list(b"ACGT"), 240, 90)
    orc = oracle.OracleIndex.from_raw(raw)
    lo, hi = shard.partition_reads(offs, world)[rank]
    s, o = shard.shard(seqs, offs, lo, hi)
    lens = orc.pml(s, o)
    f, a, b, ssum = oracle.classify(lens, o, 20, 6)
    counts = shard.allreduce_counts(shard.class_counts(a, b, int(o[-1])))
    np.save(os.path.join(tmp, f"lens{rank}.npy"), lens)
    if rank == 0:
        np.save(os.path.join(tmp, "counts.npy"), counts.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharding_matches_single_process(tmp_path, oracle_mod):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    raw, text = cases.real_case(5, 4000, list(b"ACGT"))
    rng = np.random.default_rng(9)
    seqs, offs = cases.reads_mixed(rng, text, list(b"ACGT"), 240, 90)
    orc = oracle_mod.OracleIndex.from_raw(raw)
    want = orc.pml(seqs, offs)
    got = np.concatenate([np.load(tmp_path / f"lens{r}.npy") for r in range(world)])
    assert np.array_equal(got, want)  # shards are contiguous and in order
    f, a, b, s = oracle_mod.classify(want, offs, 20, 6)
    counts = np.load(tmp_path / "counts.npy")
    assert counts.tolist() == [240, int(offs[-1]), int(f.sum()), int(240 - f.sum())]
