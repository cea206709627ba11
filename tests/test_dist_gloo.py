"""CPU suite, part 3: the N>1 path (chromosome shards + sum-reduce of the count vector) with
world_size-2 gloo process groups.  The per-rank counting call is a stand-in here (the CPU oracle);
what is under test is the sharding and the reduction -- that the reduced vector is the single-process one."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gtx import shard, synth
from oracle import orc


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, stranded, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        refs = synth.genome_intervals(4000, 51, 50, 3000, stranded=stranded)
        reads = synth.genome_intervals(50000, 52, 30, 200, stranded=stranded)
        total, n_mine = shard.sharded_count(lambda r: orc.count(refs, r, algo=orc.SORTED_MERGE), reads, len(synth.CHROM_NAMES),
                                            synth.CHROM_LEN, rank, world, dist)
        n = torch.tensor([n_mine])
        dist.all_reduce(n)
        if rank == 0:
            np.save(out, total)
            assert int(n.item()) == len(reads)        # the shards partition the reads
    finally:
        dist.destroy_process_group()


def _run(world, stranded, tmp_path):
    out = str(tmp_path / ("hits_%d_%d.npy" % (world, stranded)))
    mp.spawn(_worker, args=(world, _free_port(), stranded, out), nprocs=world, join=True)
    refs = synth.genome_intervals(4000, 51, 50, 3000, stranded=stranded)
    reads = synth.genome_intervals(50000, 52, 30, 200, stranded=stranded)
    want = orc.count(refs, reads, algo=orc.SORTED_MERGE)
    np.testing.assert_array_equal(np.load(out), want)


def test_two_ranks_ignore_strand(tmp_path):
    _run(2, False, tmp_path)


def test_two_ranks_strand_classes(tmp_path):
    _run(2, True, tmp_path)


def test_lpt_balance_for_eight_ranks():
    shards = synth.lpt_shards(synth.CHROM_LEN, 8)
    assert sorted(sum(shards, [])) == list(range(24))
    loads = [synth.CHROM_LEN[s].sum() for s in shards]
    assert max(loads) / (synth.CHROM_LEN.sum() / 8) < 1.08      # chr1 alone is 8% of the genome


def test_apportion_is_exact():
    for total in (0, 1, 999, 100_000_000):
        assert int(synth.apportion(total, synth.CHROM_LEN).sum()) == total


# ---- permutation_test: ranks take disjoint shuffle ranges, exceed-counts are summed ---------------------
def _perm_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gtx import perm
        from oracle import porc
        t = perm.PermTable.synthetic(600, 40, 20, seed=5, values="gamma")
        Y = porc.statistic(t, "sum")
        P = 90                                                   # shuffles per rank
        c = porc.count_ge(t, "sum", Y, 77, rank * P, P)          # stand-in for gtx_perm_count_ge (same first_perm / n_perm contract)
        acc = torch.from_numpy(c.view(np.int64).copy())
        dist.all_reduce(acc, op=dist.ReduceOp.SUM)
        if rank == 0:
            np.save(out, acc.numpy())
    finally:
        dist.destroy_process_group()


def test_two_ranks_split_the_shuffles(tmp_path):
    from gtx import perm
    from oracle import porc
    out = str(tmp_path / "perm_counts.npy")
    mp.spawn(_perm_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    t = perm.PermTable.synthetic(600, 40, 20, seed=5, values="gamma")
    want = porc.count_ge(t, "sum", porc.statistic(t, "sum"), 77, 0, 180)
    np.testing.assert_array_equal(np.load(out), want.view(np.int64))
