"""Data-parallel gradient exchange of the training step (RCCL all-reduce of the gradients of the network being
updated, bucketed and overlapped with the backward pass).

The reference trains under `nn.DataParallel` (train_anonymizer.py:340-344), whose backward reduces the replicas' gradients onto
GPU 0. Here: one process per GPU, and the gradients of a network live in a few flat fp32 BUCKETS -- `p.grad` of every parameter is a
view into its bucket (`prepare()`), the backward kernels' results are added into those views in place, and as soon as the last
contribution to a bucket has been flushed (`bucket_ready(i)`, called from the backward sequence stage by stage) the bucket is
all-reduced asynchronously (RCCL on its own stream) while the backward of the earlier stages keeps running. `finish()` waits for the
outstanding buckets and divides by the world size. No torch.cat, no copy-back: the reduced bucket IS the gradient storage the
optimizer reads. xGMI is point-to-point (7 links/GPU): a few tens-of-MB buckets keep every ring step bandwidth-bound.
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.distributed as dist


class GradBucketReducer:
    def __init__(self, buckets: Sequence[Sequence[torch.nn.Parameter]], group=None, all_params=None):
        """buckets: parameter lists in the order their gradients become final during the backward pass.
        all_params: every parameter of the network; any that no bucket names goes into a last bucket (reduced by `finish`)."""
        self.group = group
        self.buckets: List[List[torch.nn.Parameter]] = [[p for p in b if p.requires_grad] for b in buckets]
        self.buckets = [b for b in self.buckets if b]
        seen = set()
        for b in self.buckets:
            for p in b:
                assert id(p) not in seen, "a parameter is in two buckets"
                seen.add(id(p))
        if all_params is not None:
            rest = [p for p in all_params if p.requires_grad and id(p) not in seen]
            if rest:
                self.buckets.append(rest)
        self.excluded = set()
        self.flats: List[torch.Tensor] = []
        self.views: List[List[torch.Tensor]] = []
        self.pending = []           # (bucket index, work handle)
        self.issued: List[int] = [] # bucket indices in launch order of the current step (tests, logging)
        self.ready = set()
        self.exchange = True        # False: the buckets are not all-reduced (bench.py times a step with and without the exchange: its share)
        self.force = False          # True: the collectives are launched at world size 1 too (tests: the asynchronous RCCL path on a one-GPU box)

    def nbytes(self) -> int:
        """Bytes one step's all-reduce moves per rank (fp32 gradients of every bucket)."""
        return 4 * sum(p.numel() for b in self.buckets for p in b)

    def world(self) -> int:
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def _allocate(self):
        self.flats, self.views = [], []
        for b in self.buckets:
            n = sum(p.numel() for p in b)
            flat = torch.zeros(n, dtype=torch.float32, device=b[0].device)
            views, off = [], 0
            for p in b:
                views.append(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
            self.flats.append(flat)
            self.views.append(views)

    def prepare(self, exclude=()):
        """Start of a step (instead of optimizer.zero_grad): every p.grad becomes a zeroed view into its bucket.
        exclude: parameters that get NO gradient this step (their .grad stays None, so the optimizer skips them -- the FrozenBN
        buffers of action training); their bucket slots stay zero on every rank."""
        if not self.flats or any(f.device != b[0].device for f, b in zip(self.flats, self.buckets)):
            self._allocate()
        self.excluded = {id(p) for p in exclude}
        for flat, b, views in zip(self.flats, self.buckets, self.views):
            flat.zero_()
            for p, v in zip(b, views):
                p.grad = None if id(p) in self.excluded else v
        self.pending, self.issued, self.ready = [], [], set()

    def bucket_ready(self, i: int):
        """All contributions to bucket i are in its views (enqueued on the current stream): launch its all-reduce."""
        if i in self.ready or i >= len(self.buckets):
            return
        self.ready.add(i)
        for p, v in zip(self.buckets[i], self.views[i]):
            if id(p) in self.excluded:
                continue
            if p.grad is not v:                       # a backward step replaced the view by a fresh tensor: bring it home
                if p.grad is not None:
                    v.copy_(p.grad)
                p.grad = v
        self.issued.append(i)
        if self.exchange and (self.world() > 1 or (self.force and dist.is_initialized())):
            self.pending.append((i, dist.all_reduce(self.flats[i], op=dist.ReduceOp.SUM, group=self.group, async_op=True)))

    def finish(self):
        """Launch whatever was not announced, wait for every bucket, turn sums into means."""
        for i in range(len(self.buckets)):
            self.bucket_ready(i)
        w = self.world()
        for _, work in self.pending:
            work.wait()
        if w > 1 and self.exchange:
            torch._foreach_div_(self.flats, float(w))
        self.pending = []
