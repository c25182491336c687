"""CPU, world_size 2 over gloo: clip sharding + feature all-gather (the N > 1 path of
bench.py / extraction.extract_video_sharded; on the GPU box the same code runs over RCCL)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ted_spad_amd import sharding


def test_shard_range_partitions():
    for T in (1, 2, 7, 225, 450, 1800):
        for world in (1, 2, 3, 4, 8):
            spans = [sharding.shard_range(T, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == T
            for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
                assert a1 == b0 and a0 <= a1
            per = -(-T // world)
            assert all(hi - lo <= per for lo, hi in spans)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, T, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.arange(T * 3 * 8, dtype=torch.float32).view(T, 3, 8)
        lo, hi = sharding.shard_range(T, rank, world)
        got = sharding.gather_video_features(full[lo:hi].clone(), T)
        q.put((rank, bool(torch.equal(got, full)), tuple(got.shape)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("T", [10, 7, 1])
def test_gather_video_features_world2(T):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, T, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, shape in res:
        assert ok and shape == (T, 3, 8), (rank, ok, shape)


def test_world1_passthrough():
    x = torch.randn(5, 2, 4)
    assert sharding.gather_video_features(x, 5) is x


def _grad_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ted_spad_amd.train_step import allreduce_mean_grads
        ps = [torch.nn.Parameter(torch.zeros(3, 4)), torch.nn.Parameter(torch.zeros(5)), torch.nn.Parameter(torch.zeros(2))]
        ps[0].grad = torch.full((3, 4), float(rank + 1))
        ps[1].grad = torch.arange(5, dtype=torch.float32) * (rank + 1)
        # ps[2] has no gradient on any rank (a frozen parameter): skipped consistently
        allreduce_mean_grads(ps)
        ok = bool(torch.allclose(ps[0].grad, torch.full((3, 4), 1.5)) and torch.allclose(ps[1].grad, torch.arange(5.) * 1.5) and ps[2].grad is None)
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_gradient_allreduce_world2():
    """The data-parallel exchange of the training step (one flat all-reduce of the updated net's grads)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)
