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


# ---- bucketed gradient all-reduce (grad_reduce.GradBucketReducer) ------------------------------------------------------

def _bucket_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ted_spad_amd.grad_reduce import GradBucketReducer
        from ted_spad_amd.train_step import allreduce_mean_grads
        g = torch.Generator().manual_seed(100 + rank)
        shapes = [(7, 3), (5,), (2, 2, 2), (11,), (4, 4), (3,)]
        grads = [torch.randn(s, generator=g) for s in shapes]
        # reference: the flat exchange after the whole backward
        pa = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
        for p, gr in zip(pa, grads):
            p.grad = gr.clone()
        allreduce_mean_grads(pa)
        # bucketed: three buckets finished one after the other by a mock backward; one parameter left to the `rest` bucket, one frozen
        pb = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
        frozen = torch.nn.Parameter(torch.zeros(3))
        red = GradBucketReducer([[pb[0], pb[1]], [pb[2]], [pb[3], pb[4]]], all_params=pb + [frozen])
        red.prepare(exclude=[frozen])
        issued_during_backward = []
        for bi, members in enumerate(((0, 1), (2,), (3, 4))):
            for m in members:
                pb[m].grad.add_(grads[m])                 # the backward kernels add into the bucket views in place
            red.bucket_ready(bi)
            issued_during_backward.append(len(red.issued))
        pb[5].grad.add_(grads[5])                         # a parameter no bucket names: reduced by finish()
        red.finish()
        same = all(torch.equal(a.grad, b.grad) for a, b in zip(pa, pb))
        views = all(b.grad.untyped_storage().data_ptr() == red.flats[i].untyped_storage().data_ptr()
                    for i, bucket in enumerate(red.buckets) for b in bucket if b is not frozen)
        q.put((rank, same, views, issued_during_backward, list(red.issued), frozen.grad is None))
    finally:
        dist.destroy_process_group()


def test_bucketed_gradient_allreduce_matches_flat_world2():
    """Bit-equal to the flat all-reduce; every bucket is launched as soon as its stage is done (before the backward ends); the
    optimizer's gradients ARE the bucket storage (no copy-back); excluded parameters keep grad None."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same, views, during, issued, frozen_none in res:
        assert same and views and frozen_none, (rank, same, views, frozen_none)
        assert during == [1, 2, 3] and issued == [0, 1, 2, 3]


class _StubExtractor:
    """Host-logic stand-in for the feature extractor (NOT a compute fallback: tests of the sharding / collective only)."""
    feature_dim = 6

    def extract_features(self, x):
        return x.flatten(1).mean(1, keepdim=True).repeat(1, 6).view(-1, 6, 1, 1, 1)


def _short_video_worker(rank, world, port, T, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ted_spad_amd import extraction
        ncrops = 2
        clips = torch.arange(T * ncrops, dtype=torch.float32).view(T * ncrops, 1, 1, 1, 1).expand(T * ncrops, 3, 2, 4, 4).contiguous()
        lo, hi = sharding.shard_range(T, rank, world)
        full = extraction.extract_video_sharded(_StubExtractor(), clips[lo * ncrops:hi * ncrops], T, ncrops=ncrops, batch=3)
        want = torch.arange(T * ncrops, dtype=torch.float32).view(T, ncrops, 1).expand(T, ncrops, 6)
        q.put((rank, tuple(full.shape), bool(torch.equal(full, want)), hi - lo))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("T", [1, 3])
def test_extract_video_sharded_with_empty_shards(T):
    """T < world (or T = 3 on 2 ranks: blocks of 2 and 1): an empty shard launches nothing, takes its feature width from the model,
    and still joins the all-gather (the probe forward crashed on n == 0 and the other ranks hung)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_short_video_worker, args=(r, 2, port, T, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, shape, ok, n in res:
        assert shape == (T, 2, 6) and ok, (rank, shape, ok)
    if T == 1:
        assert sorted(n for _, _, _, n in res) == [0, 1]


class _StubAnonymizer:
    """Host-logic stand-in for fa (NOT a compute fallback): a per-frame function that depends on the colour channel, so that the Q1 reshape feed and the
    permute feed give different clips."""

    def __call__(self, frames):                 # (n, 3, H, W)
        return frames * torch.tensor([1.0, 2.0, 3.0]).view(1, 3, 1, 1) + 0.5


class _StubExtractorCT:
    feature_dim = 4

    def extract_features(self, x):              # (b, 3, T, H, W): a function that sees WHICH (channel, frame) slot a value sits in
        w = torch.arange(1, 3 * x.shape[2] + 1, dtype=torch.float32).view(1, 3, x.shape[2], 1, 1)
        return (x * w).flatten(1).sum(1, keepdim=True).repeat(1, 4).view(-1, 4, 1, 1, 1)


def _anon_worker(rank, world, port, T, layout, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ted_spad_amd import extraction
        ncrops = 2
        g = torch.Generator().manual_seed(7)
        clips = torch.rand((T * ncrops, 4, 3, 2, 2), generator=g)                      # loader layout (n, frames, 3, H, W)
        lo, hi = sharding.shard_range(T, rank, world)
        fa, ft = _StubAnonymizer(), _StubExtractorCT()
        full = extraction.extract_video_sharded(ft, clips[lo * ncrops:hi * ncrops], T, ncrops=ncrops, fa_model=fa, layout=layout, fa_batch=3)
        # the single-process answer: every clip through extraction.feed (the reference's fa + reshape feed) and the extractor
        want = ft.extract_features(extraction.feed(clips, fa, layout)).flatten(1).view(T, ncrops, 4)
        other = ft.extract_features(extraction.feed(clips, fa, "permute" if layout == "reference" else "reference")).flatten(1).view(T, ncrops, 4)
        q.put((rank, tuple(full.shape), bool(torch.allclose(full, want)), bool(torch.allclose(full, other))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("layout", ["reference", "permute"])
@pytest.mark.parametrize("T", [5, 1])
def test_extract_video_sharded_with_the_anonymizer(T, layout):
    """The sharded path runs what the reference's extractors actually do (anonymized = True hard-coded, dali_extraction.py:108,169-178): clip -> fa -> Q1 feed ->
    ft.extract_features -> all-gather; world 2, ragged and empty shards; the gathered block equals the single-process result and the two feeds differ."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_anon_worker, args=(r, 2, port, T, layout, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, shape, ok, same_as_other in res:
        assert shape == (T, 2, 4) and ok and not same_as_other, (rank, shape, ok, same_as_other)
