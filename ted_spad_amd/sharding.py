"""Clip sharding of one long video across the GPUs of a node + reassembly of the per-video
feature tensor with ONE collective (RCCL all-gather over xGMI; `nccl` backend == RCCL on ROCm).

New design, not a translation: the reference's multi-GPU extraction is nn.DataParallel and is
effectively broken (dali_extraction.py:33,128-133,175-178 -- SURVEY.md §2.1). Clips of a video are
independent units (dali_extraction.py:62-73: one 16-frame clip per 32 source frames), so rank r
takes a contiguous block of ceil(T/P) clip times (row order of the .npy stays trivial) and the
only exchange is the (T_r, ncrops, F) fp32 feature block: ~1.8-18 MB per video.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(T: int, rank: int, world: int):
    """Contiguous block [lo, hi) of clip times for `rank`; blocks differ by at most `per` rows,
    trailing ranks may be empty when T < world."""
    per = -(-T // world)
    lo = min(rank * per, T)
    return lo, min(lo + per, T)


def gather_video_features(local: torch.Tensor, T: int, group=None) -> torch.Tensor:
    """local: (T_r, ...) feature rows of this rank's block (shard_range order).
    Returns the full (T, ...) tensor on every rank. One all_gather_into_tensor of equal-size
    (padded) blocks: on the full xGMI mesh each rank exchanges its block with all peers at once."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        assert local.shape[0] == T
        return local
    rank = dist.get_rank(group)
    per = -(-T // world)
    lo, hi = shard_range(T, rank, world)
    assert local.shape[0] == hi - lo, "rank %d holds %d rows, expected %d" % (rank, local.shape[0], hi - lo)
    tail = tuple(local.shape[1:])
    send = local.new_zeros((per,) + tail)
    send[: hi - lo] = local
    recv = local.new_empty((world * per,) + tail)
    dist.all_gather_into_tensor(recv, send.contiguous(), group=group)
    return recv[:T]


def batch_plan(n_local: int, batch: int, streams: int = 1, min_batch: int = 32):
    """[(first clip, clips)] forwards for a shard of `n_local` clips: as few forwards of at most `batch` clips as cover the shard -- but at least one per HIP
    stream as long as a forward keeps `min_batch` clips -- and the clips spread evenly over them. One video split over 8 ranks (cfg4: 29 clip times x 10 crops = 290 clips per rank) becomes 145 + 145 on two streams instead of
    one ragged 290-clip forward on one stream; 2 250 clips at 375 per forward stay 6 x 375."""
    if n_local <= 0:
        return []
    nb = -(-n_local // batch)
    if nb < streams and n_local // streams >= min_batch:
        nb = streams
    per = -(-n_local // nb)
    return [(i, min(per, n_local - i)) for i in range(0, n_local, per)]
