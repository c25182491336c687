"""The three losses of the anonymizer training step on MI355X, behind the reference's own
class interfaces. Each forward is ONE fused HIP launch that also produces the input
gradients, which `backward` scales by the incoming gradient.

    NTXentLoss(device, batch_size, temperature, use_cosine_similarity)(zis, zjs)   aux_code/nt_xent_original.py:7-70
    TripletMarginLoss(margin=1.0)(anchor, positive, negative)                       train_anonymizer.py:349-350,115
    CrossEntropyLoss()(logits, labels)                                              train_anonymizer.py:347,107

The reference re-instantiates NTXentLoss (and rebuilds its numpy mask) every iteration
(train_anonymizer.py:82,155); here the mask is in-kernel.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from .engine import _stream_ptr, require_cuda


def _f32(t):
    return t.contiguous().float()


class _NTXentFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, zis, zjs, temperature, cosine):
        require_cuda(zis, "NTXentLoss")
        zi, zj = _f32(zis), _f32(zjs)
        n, c = zi.shape
        loss = torch.empty(1, dtype=torch.float32, device=zi.device)
        dzi, dzj = torch.empty_like(zi), torch.empty_like(zj)
        _lib.check(_lib.lib().tedspad_ntxent_fwd_bwd(zi.data_ptr(), zj.data_ptr(), loss.data_ptr(), dzi.data_ptr(), dzj.data_ptr(),
                                                     n, c, C.c_float(temperature), int(cosine), _stream_ptr()), "tedspad_ntxent_fwd_bwd")
        ctx.save_for_backward(dzi, dzj)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        dzi, dzj = ctx.saved_tensors
        return dzi * g, dzj * g, None, None


class NTXentLoss(nn.Module):
    def __init__(self, device, batch_size, temperature, use_cosine_similarity):
        super().__init__()
        self.batch_size, self.temperature, self.device = batch_size, temperature, device
        self.use_cosine_similarity = bool(use_cosine_similarity)

    def forward(self, zis, zjs):
        if zis.shape[0] != self.batch_size or zjs.shape != zis.shape:
            raise ValueError("NTXentLoss built for batch_size=%d got %s / %s" % (self.batch_size, tuple(zis.shape), tuple(zjs.shape)))
        return _NTXentFn.apply(zis, zjs, float(self.temperature), self.use_cosine_similarity)


class _TripletFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, p, n, margin, eps):
        require_cuda(a, "TripletMarginLoss")
        a, p, n = _f32(a), _f32(p), _f32(n)
        b, c = a.shape
        loss = torch.empty(1, dtype=torch.float32, device=a.device)
        ws = torch.empty(b, dtype=torch.float32, device=a.device)
        da, dp, dn = torch.empty_like(a), torch.empty_like(a), torch.empty_like(a)
        _lib.check(_lib.lib().tedspad_triplet_fwd_bwd(a.data_ptr(), p.data_ptr(), n.data_ptr(), loss.data_ptr(), ws.data_ptr(),
                                                      da.data_ptr(), dp.data_ptr(), dn.data_ptr(), b, c, C.c_float(margin),
                                                      C.c_float(eps), _stream_ptr()), "tedspad_triplet_fwd_bwd")
        ctx.save_for_backward(da, dp, dn)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        da, dp, dn = ctx.saved_tensors
        return da * g, dp * g, dn * g, None, None


class TripletMarginLoss(nn.Module):
    def __init__(self, margin=1.0, p=2.0, eps=1e-6):
        super().__init__()
        if p != 2.0:
            raise NotImplementedError("only p=2 (the reference's default)")
        self.margin, self.eps = margin, eps

    def forward(self, anchor, positive, negative):
        return _TripletFn.apply(anchor, positive, negative, float(self.margin), float(self.eps))


class _CEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels):
        require_cuda(logits, "CrossEntropyLoss")
        lg = _f32(logits)
        lab = labels.contiguous().long()
        b, c = lg.shape
        loss = torch.empty(1, dtype=torch.float32, device=lg.device)
        ws = torch.empty(b, dtype=torch.float32, device=lg.device)
        dl = torch.empty_like(lg)
        _lib.check(_lib.lib().tedspad_cross_entropy_fwd_bwd(lg.data_ptr(), lab.data_ptr(), loss.data_ptr(), ws.data_ptr(),
                                                            dl.data_ptr(), b, c, _stream_ptr()), "tedspad_cross_entropy_fwd_bwd")
        ctx.save_for_backward(dl)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None


class CrossEntropyLoss(nn.Module):
    def forward(self, logits, labels):
        return _CEFn.apply(logits, labels)
