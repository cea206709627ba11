"""Multi-GPU sharding of the count path: chromosomes are dealt to the ranks, every rank
counts its own reads against the replicated reference set, and the per-region count vectors are summed
(RCCL all-reduce over xGMI on GPUs).  Two regions can only overlap inside one chromosome
(genomic_intervals.cpp:624-630), so the per-rank vectors have disjoint non-zero entries and the sum is
the single-device result bit for bit.

`count_fn(reads) -> int64/uint64 vector` is the per-rank counting call; on the GPU it is
gtx.Engine.count_device into a torch tensor, in the CPU (gloo) tests it is whatever the test supplies.
"""
import numpy as np

from . import synth


def chrom_of_class(cls, n_chrom):
    """class ids fold the strand as strand * n_chrom + chrom rank"""
    return np.asarray(cls) % n_chrom


def rank_chroms(weights, world):
    """LPT assignment: chromosome indices owned by each rank."""
    return synth.lpt_shards(weights, world)


def shard_reads(reads, n_chrom, owned):
    """The reads of one rank: those on the chromosomes it owns (order preserved)."""
    mask = np.isin(chrom_of_class(reads[:, 0], n_chrom), np.asarray(owned, dtype=np.int64))
    return reads[mask]


def reduce_counts(local_hits, dist=None, device_tensor=False):
    """Sum of the per-rank count vectors over the process group (identity for a single process).

    uint64 counts travel as int64: two's-complement addition is the same bit pattern."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local_hits
    import torch
    if device_tensor:
        dist.all_reduce(local_hits, op=dist.ReduceOp.SUM)
        return local_hits
    t = torch.from_numpy(np.ascontiguousarray(local_hits).view(np.int64).copy())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.numpy().view(np.uint64)


def sharded_count(count_fn, reads, n_chrom, chrom_weights, rank, world, dist=None):
    """Whole multi-rank count of `reads` as seen from `rank`: shard, count locally, reduce."""
    owned = rank_chroms(chrom_weights, world)[rank]
    mine = shard_reads(reads, n_chrom, owned)
    return reduce_counts(count_fn(mine), dist), len(mine)
