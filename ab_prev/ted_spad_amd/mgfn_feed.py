"""Device-side MGFN feature feed: what `Dataset.__getitem__`
(anomaly_detection_mgfn/datasets/dataset.py:51-100) does to a loaded `.npy` -- `process_feat` 32-segment mean pooling
(utils/utils.py:34-42) per crop plus the L2-magnitude channel -- on features that are still in HBM after
extraction, in one kernel (tedspad_segment_pool_mag)."""
from __future__ import annotations

import torch

from . import _lib
from ._lib import check
from .engine import _stream_ptr, require_cuda


def _launch(feat: torch.Tensor, length: int) -> torch.Tensor:
    require_cuda(feat, "mgfn_feed")
    if feat.dim() == 2:
        feat = feat.unsqueeze(1)                      # dataset.py:67-68: (T,F) -> (T,1,F)
    if feat.dim() != 3:
        raise ValueError("mgfn_feed: features must be (T,F) or (T,ncrops,F)")
    feat = feat.to(torch.float32).contiguous()        # dataset.py:54: np.array(features, dtype=np.float32)
    t, nc, f = feat.shape
    out = torch.empty((nc, length, f + 1) if length > 0 else (t, nc, f + 1), dtype=torch.float32, device=feat.device)
    check(_lib.lib().tedspad_segment_pool_mag(feat.data_ptr(), t, nc, f, int(length), out.data_ptr(), _stream_ptr()), "tedspad_segment_pool_mag")
    return out


def getitem(features: torch.Tensor, test_mode: bool = False, seg_length: int = 32) -> torch.Tensor:
    """features: (T,F) or (T,ncrops,F) on the GPU. Train: (ncrops, seg_length, F+1); test_mode: (T, ncrops, F+1)."""
    if not test_mode and seg_length <= 0:
        raise ValueError("seg_length must be positive")
    return _launch(features, 0 if test_mode else seg_length)


def process_feat(feat: torch.Tensor, length: int) -> torch.Tensor:
    """utils/utils.py:34-42 for one (T,F) matrix -> (length, F)."""
    if feat.dim() != 2:
        raise ValueError("process_feat: feat must be (T,F)")
    return _launch(feat, length)[0, :, :-1]
