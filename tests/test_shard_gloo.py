"""world_size-2 (and 3) gloo tests of the N>1 path: contiguous batch shards, no data-path
collective, all-gather of result term counts only."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from csgn_amd.shard import gather_term_counts, owner_of, product_term_counts, shard_range


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("total,world", [(8, 2), (7, 2), (1, 2), (0, 2), (65536, 8), (10, 3), (1048576, 8), (5, 8)])
def test_shard_ranges_partition_the_batch(total, world):
    ranges = [shard_range(total, r, world) for r in range(world)]
    assert ranges[0][0] == 0 and ranges[-1][1] == total
    for (a, b), (c, d) in zip(ranges, ranges[1:]):
        assert b == c and a <= b
    for r, (lo, hi) in enumerate(ranges):
        for p in {lo, hi - 1} if hi > lo else set():
            assert owner_of(p, total, world) == r


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # a global ragged batch, identical on every rank (seeded), sharded by contiguous range
        g = torch.Generator().manual_seed(1234)
        t1 = torch.randint(0, 9, (total,), generator=g)
        t2 = torch.randint(0, 9, (total,), generator=g)
        offl = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(t1, 0)])
        offr = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(t2, 0)])
        lo, hi = shard_range(total, rank, world)
        local = product_term_counts(offl[lo:hi + 1], offr[lo:hi + 1])
        assert local.numel() == hi - lo
        allc = gather_term_counts(local, total)
        want = t1 * t2
        ok = torch.equal(allc, want)
        # barrier + max-over-ranks timing idiom used by bench.py
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        q.put((rank, ok, float(t.item()), int(allc.sum())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,total", [(2, 64), (2, 7), (3, 10)])
def test_gather_term_counts_gloo(world, total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in results) == list(range(world))
    assert all(r[1] for r in results), "gathered term counts differ from the single-process answer"
    assert all(r[2] == float(world) for r in results)
    assert len({r[3] for r in results}) == 1          # every rank holds the same gathered vector
