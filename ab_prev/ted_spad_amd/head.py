"""fp32 head ops (Linear [+ BatchNorm1d eval] [+ ReLU], row L2-normalise) -> libtedspad_hip.so.
Reference: I3Res50.fc (large_i3d.py:147,245), mlp.forward (model_loaders.py:250-254)."""
import ctypes as C

import torch

from . import _lib
from .engine import _stream_ptr, require_cuda


def linear(x, weight, bias=None, bn=None, relu=False):
    """x (B,K) fp32 cuda -> (B,N). `bn` = BNParams in eval mode, folded into scale/shift."""
    require_cuda(x, "head.linear")
    x = x.contiguous().float()
    w = weight.detach().contiguous().float()
    B, K = x.shape
    N = w.shape[0]
    scale = shift = None
    if bn is not None:
        from .engine import fold_bn
        scale, shift = fold_bn(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, conv_bias=bias)
    elif bias is not None:
        shift = bias.detach().float().contiguous()
    y = torch.empty((B, N), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().tedspad_linear_fwd(x.data_ptr(), w.data_ptr(), scale.data_ptr() if scale is not None else None,
                                             shift.data_ptr() if shift is not None else None, y.data_ptr(), B, K, N,
                                             int(relu), _stream_ptr()), "tedspad_linear_fwd")
    return y


def l2_normalize(x, eps=1e-12):
    require_cuda(x, "head.l2_normalize")
    x = x.contiguous().float()
    y = torch.empty_like(x)
    _lib.check(_lib.lib().tedspad_l2_normalize_rows(x.data_ptr(), y.data_ptr(), x.shape[0], x.shape[1], C.c_float(eps),
                                                    _stream_ptr()), "tedspad_l2_normalize_rows")
    return y
