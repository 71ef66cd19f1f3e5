"""Multi-GPU layout on CPU: LPT sharding is a partition; the weight broadcast works over gloo
with world_size 2 (the RCCL path uses the same code with backend 'nccl')."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from kokoro_align_amd.sharding import lattice_cost, lpt_partition, shard_for_rank


def test_lpt_partition_is_a_balanced_partition():
    costs = [lattice_cost(t, s) for t, s in [(50000, 5000), (81140, 2000), (20000, 300), (160000, 22000),
                                             (30000, 4000), (30000, 4000), (1000, 10), (99999, 14000)]]
    for n in (1, 2, 3, 8):
        parts = lpt_partition(costs, n)
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(len(costs)))
        loads = [sum(costs[i] for i in p) for p in parts]
        assert max(loads) - min(loads) <= max(costs)
    assert lpt_partition([], 4) == [[], [], [], []]
    assert lattice_cost(100, 3, 1000) == 100 * 7


def test_shard_for_rank_covers_everything():
    shapes = [(1000 * (i + 1), 100 * (i + 1)) for i in range(13)]
    seen = sorted(i for r in range(4) for i in shard_for_rank(shapes, r, 4))
    assert seen == list(range(13))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kokoro_align_amd.sharding import broadcast_model_weights, gather_rank_stats
        model = broadcast_model_weights(torch.device("cpu"))
        n = sum(p.numel() for p in model.parameters())
        digest = float(sum(p.double().sum() for p in model.parameters()))
        shapes = [(1000 * (i + 1), 100 * (i + 1)) for i in range(9)]
        mine = shard_for_rank(shapes, rank, world)
        stats = gather_rank_stats(sum(shapes[i][0] for i in mine), 1.0 + rank, torch.device("cpu"))
        q.put((rank, n, digest, mine, stats))
    finally:
        dist.destroy_process_group()


def test_weight_broadcast_and_sharding_gloo_ws2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, n0, d0, mine0, st0), (r1, n1, d1, mine1, st1) = res
    assert n0 == n1 == 579367                     # AudioToChar parameter count
    assert d0 == d1                               # identical weights after the broadcast
    assert sorted(mine0 + mine1) == list(range(9)) and not set(mine0) & set(mine1)
    assert st0 == st1 and sum(f for f, _ in st0) == sum(1000 * (i + 1) for i in range(9))
