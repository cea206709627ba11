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


# ---- one global read set, sharded by the product's LPT packing (bench.py N > 1, genomic_overlaps --ngpu) ------------------
def test_product_lpt_packing():
    import gtx
    for world in (1, 2, 3, 4, 8):
        per_chrom = synth.apportion(100_000_000 * world, synth.CHROM_LEN)
        owner = gtx.lpt_assign(per_chrom, world)
        assert owner.min() >= 0 and owner.max() < world and len(owner) == 24
        loads = np.bincount(owner, weights=per_chrom, minlength=world)
        assert loads.min() > 0
        assert loads.max() / loads.mean() < 1.08                      # chr1 alone is 8 % of the genome: 8 GPUs cannot do better than ~1.04
        np.testing.assert_array_equal(owner, gtx.lpt_assign(per_chrom, world))     # deterministic
    # zero and equal loads: every class still gets an owner, ties go to the lowest member
    assert gtx.lpt_assign([5, 5, 5, 5], 2).tolist() == [0, 1, 0, 1]
    assert gtx.lpt_assign([0, 0, 7], 2).tolist() == [1, 0, 0] or gtx.lpt_assign([0, 0, 7], 2).max() < 2


def _reads_of(chroms, per, seed):
    parts = []
    for c, k in zip(chroms, per):
        rng = np.random.default_rng(seed * 1000 + int(c))
        s = np.sort(rng.integers(1, int(synth.CHROM_LEN[c]) - 60, size=int(k)))
        parts.append(np.stack([np.full(int(k), c), s, s + 49], axis=1))
    return np.concatenate(parts).astype(np.int32) if parts else np.zeros((0, 3), dtype=np.int32)


def _strong_worker(rank, world, port, total, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gtx
        refs = synth.genome_intervals(3000, 71, 50, 3000)
        per_chrom = synth.apportion(total, synth.CHROM_LEN)
        owner = gtx.lpt_assign(per_chrom, world)
        mine = np.nonzero(owner == rank)[0]
        reads = _reads_of(mine, per_chrom[mine], 9)
        hits = orc.count(refs, reads, algo=orc.SORTED_MERGE)           # stand-in for gtx_count_device on this rank's GPU
        t = torch.from_numpy(hits.view(np.int64).copy())
        dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)                    # the exchange step: reduce(sum) of the count vector to rank 0
        n = torch.tensor([len(reads)])
        dist.all_reduce(n)
        if rank == 0:
            assert int(n.item()) == total                              # ONE fixed read set, partitioned by the ranks
            np.save(out, t.numpy())
    finally:
        dist.destroy_process_group()


def test_two_ranks_share_one_fixed_read_set(tmp_path):
    total = 60_000
    out = str(tmp_path / "strong.npy")
    mp.spawn(_strong_worker, args=(2, _free_port(), total, out), nprocs=2, join=True)
    refs = synth.genome_intervals(3000, 71, 50, 3000)
    per_chrom = synth.apportion(total, synth.CHROM_LEN)
    reads = _reads_of(np.arange(24), per_chrom, 9)
    want = orc.count(refs, reads, algo=orc.SORTED_MERGE)
    np.testing.assert_array_equal(np.load(out).view(np.uint64), want)


# ---- round 3: the product's own N > 1 scheme -- every member finalizes its classes into its piece of a compact vector
# (gtx_group_plan), the pieces travel to member 0, member 0 puts them into file order.  Here: the planning function of libgtx.so
# (pure host code) in two gloo ranks, the per-rank counting by the CPU oracle, the pieces moved with gloo send / recv.
def _piece_worker(rank, world, port, out):
    import gtx
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(9)
        refs = synth.genome_intervals(5000, 61, 50, 3000)
        refs = refs[rng.permutation(len(refs))]                          # file order is not class order
        reads = synth.genome_intervals(60000, 62, 30, 200)
        owner = gtx.lpt_assign(np.bincount(reads[:, 0], minlength=24), world)     # deterministic: every rank computes the same
        seg, perm = gtx.group_plan(refs[:, 0], owner, world)
        mine = reads[owner[reads[:, 0]] == rank]
        full = orc.count(refs, mine, algo=orc.BIN_INDEX)                 # this rank's reads only
        piece = torch.from_numpy(full[perm[seg[rank]:seg[rank + 1]]].astype(np.int64))    # ... and only its own regions travel
        assert int(full.sum()) == int(piece.sum())                       # its reads hit no region of another member's classes
        if rank == 0:
            compact = torch.zeros(len(refs), dtype=torch.int64)
            compact[seg[0]:seg[1]] = piece
            for r in range(1, world):
                buf = torch.zeros(int(seg[r + 1] - seg[r]), dtype=torch.int64)
                dist.recv(buf, src=r)
                compact[seg[r]:seg[r + 1]] = buf
            hits = np.zeros(len(refs), dtype=np.uint64)
            hits[perm] = compact.numpy().view(np.uint64)
            np.save(out, hits)
        else:
            dist.send(piece, dst=0)
    finally:
        dist.destroy_process_group()


def test_pieces_of_the_compact_vector_two_and_three_ranks(tmp_path):
    rng = np.random.default_rng(9)
    refs = synth.genome_intervals(5000, 61, 50, 3000)
    refs = refs[rng.permutation(len(refs))]
    reads = synth.genome_intervals(60000, 62, 30, 200)
    want = orc.count(refs, reads, algo=orc.BIN_INDEX)
    for world in (2, 3):
        out = str(tmp_path / ("pieces_%d.npy" % world))
        mp.spawn(_piece_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        np.testing.assert_array_equal(np.load(out), want)


def test_group_plan_is_a_permutation_by_owner():
    import gtx
    rng = np.random.default_rng(4)
    cls = rng.integers(-1, 30, size=5000).astype(np.int32)               # -1: placeholders; 24..29: beyond the assignment
    owner = rng.integers(0, 5, size=24).astype(np.int32)
    seg, perm = gtx.group_plan(cls, owner, 5)
    assert seg[0] == 0 and seg[-1] == len(cls) and np.all(np.diff(seg) >= 0)
    assert np.array_equal(np.sort(perm), np.arange(len(cls)))
    own = np.where((cls >= 0) & (cls < 24), owner[np.clip(cls, 0, 23)], 0)
    for m in range(5):
        piece = perm[seg[m]:seg[m + 1]]
        assert np.all(own[piece] == m) and np.all(np.diff(piece) > 0)        # its regions, in file order
