"""ORACLE (test infrastructure, never shipped): fp32 CPU reference of the fused
conv + scale/shift (+residual) + ReLU/sigmoid op in channels-last form, built on
torch.nn.functional.conv3d -- the reference's own arithmetic (its Conv3d/Conv2d modules:
aux_code/models/i3d.py:72-77, large_i3d.py:47-51, unet_parts.py:16-19).
Only tests/, smoke() and bench.py's cpu_baseline leg may import this."""
import torch
import torch.nn.functional as F


def conv_cl(x, w, scale, shift, stride=(1, 1, 1), pads=(0, 0, 0), pads_back=None, residual=None, relu=True, sigmoid=False):
    """x: (n,t,h,w,cin) fp32; w: (cout,cin,kt,kh,kw) fp32; returns (n,to,ho,wo,cout) fp32."""
    pb = pads if pads_back is None else pads_back
    xi = x.permute(0, 4, 1, 2, 3)
    xi = F.pad(xi, (pads[2], pb[2], pads[1], pb[1], pads[0], pb[0]))
    y = F.conv3d(xi, w, stride=stride).permute(0, 2, 3, 4, 1)
    y = y * scale + shift
    if residual is not None:
        y = y + residual
    if relu:
        y = F.relu(y)
    if sigmoid:
        y = torch.sigmoid(y)
    return y


def maxpool_cl(x, k, s, pads=(0, 0, 0), pads_back=None, pad_zero=False):
    pb = pads if pads_back is None else pads_back
    xi = x.permute(0, 4, 1, 2, 3)
    xi = F.pad(xi, (pads[2], pb[2], pads[1], pb[1], pads[0], pb[0]), value=0.0 if pad_zero else float("-inf"))
    return F.max_pool3d(xi, k, s).permute(0, 2, 3, 4, 1)
