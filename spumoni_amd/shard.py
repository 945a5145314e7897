"""Sharding of a read set across the GPUs of one node.

Reads are independent queries on a read-only index (src/compute_ms_pml.cpp:907-938
carries no cross-read state), so the index is replicated in every GPU's HBM and
the reads are cut into contiguous, character-balanced shards, one per rank.  No
collective sits on the data path; the only exchange is the final sum of the four
classification counters {reads, bases, FOUND, NOT_PRESENT} (RCCL all-reduce on
GPUs, gloo in the CPU tests).
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np
import torch


def partition_reads(offs, world: int) -> List[Tuple[int, int]]:
    """Cut reads [0, nreads) into `world` contiguous ranges with ~equal character counts.

    offs: nreads+1 offsets (numpy / torch / list).  Ranges may be empty when there are
    fewer reads than ranks; their union is exactly [0, nreads), in order.
    """
    o = np.asarray(offs.cpu() if hasattr(offs, "cpu") else offs, dtype=np.int64)
    nreads = o.size - 1
    total = int(o[-1]) if nreads > 0 else 0
    cuts = [0]
    for w in range(1, world):
        target = total * w // world
        # first read whose start offset is >= target (never before the previous cut)
        c = int(np.searchsorted(o[:-1], target, side="left")) if nreads > 0 else 0
        cuts.append(max(cuts[-1], min(c, nreads)))
    cuts.append(nreads)
    return [(cuts[i], cuts[i + 1]) for i in range(world)]


def shard(seqs, offs, lo: int, hi: int):
    """Sub-batch of reads [lo, hi): (seqs slice, offsets rebased to 0)."""
    a, b = int(offs[lo]), int(offs[hi])
    return seqs[a:b], offs[lo : hi + 1] - offs[lo]


def class_counts(above, below, nbases: int) -> torch.Tensor:
    """[reads, bases, FOUND, NOT_PRESENT] of one shard; FOUND iff above/(above+below) > 0.5
    (compute_ms_pml.cpp:993).  Stays on the device of `above` (no host round trip: the caller may be inside a timed
    region)."""
    above = torch.as_tensor(above).to(torch.int64)
    below = torch.as_tensor(below).to(torch.int64)
    n = int(above.numel())
    found = (2 * above > above + below).sum().to(torch.int64).reshape(1)
    fixed = torch.tensor([n, int(nbases)], dtype=torch.int64, device=found.device)
    return torch.cat([fixed, found, n - found])


def allreduce_counts(counts: torch.Tensor) -> torch.Tensor:
    """Sum the four counters over all ranks (no-op without an initialised process group)."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    return counts
