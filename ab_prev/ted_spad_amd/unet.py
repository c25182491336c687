"""2-D UNet anonymizer `fa` (arch='unet') on MI355X.

Mirrors the reference `UNet` (aux_code/models/unet_model.py:6-37, unet_parts.py:8-77):
`UNet(n_channels, n_classes, bilinear=True)`, the same `state_dict` key names
(`inc.double_conv.0.weight`, `down1.maxpool_conv.1.double_conv.4.running_var`,
`up1.conv.double_conv.0.weight`, `outc.conv.bias`, ...), `forward((N,3,H,W)) -> (N,3,H,W)`
in (0,1).

Launch plan: every conv3x3+bias+BN+ReLU is one fused implicit-GEMM launch (2-D = kt 1);
each encoder level writes its skip tensor directly into the LEFT channel slice of the
decoder's concat buffer and the bilinear x2 upsample writes the RIGHT slice, so
`torch.cat([x2, x1])` (unet_parts.py:67) costs nothing; the sigmoid is the last conv's epilogue.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import engine as E
from .params import BNParams, ConvParams, params_signature


class DoubleConv(nn.Module):
    def __init__(self, cin, cout, mid=None):
        super().__init__()
        mid = mid or cout
        # indices 0,1,3,4 carry parameters exactly like the reference nn.Sequential (2 and 5 are ReLUs)
        self.double_conv = nn.Sequential(ConvParams(cin, mid, (3, 3), bias=True), BNParams(mid), nn.Identity(),
                                         ConvParams(mid, cout, (3, 3), bias=True), BNParams(cout), nn.Identity())
        self.cout = cout


class Down(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.maxpool_conv = nn.Sequential(nn.Identity(), DoubleConv(cin, cout))


class Up(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = DoubleConv(cin, cout, cin // 2)


class OutConv(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = ConvParams(cin, cout, (1, 1), bias=True)


class UNet(nn.Module):
    def __init__(self, n_channels, n_classes, bilinear=True, dtype=E.DEFAULT_DTYPE):
        super().__init__()
        if not bilinear:
            raise NotImplementedError("the reference always builds UNet with bilinear=True (model_loaders.py:32)")
        if n_channels > 8:
            raise NotImplementedError("n_channels > 8")
        self.n_channels, self.n_classes, self.bilinear = n_channels, n_classes, bilinear
        self.inc = DoubleConv(n_channels, 64)
        self.down1 = Down(64, 128)
        self.down2 = Down(128, 256)
        self.down3 = Down(256, 512)
        self.down4 = Down(512, 512)
        self.up1 = Up(1024, 256)
        self.up2 = Up(512, 128)
        self.up3 = Up(256, 64)
        self.up4 = Up(128, 64)
        self.outc = OutConv(64, n_classes)
        self.compute_dtype = dtype
        self._packed = None
        self._packed_sig = None

    def _pack_dc(self, dc: DoubleConv, dev):
        out = []
        for ci, bi in ((0, 1), (3, 4)):
            conv, bn = dc.double_conv[ci], dc.double_conv[bi]
            s, b = E.fold_bn(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, conv_bias=conv.bias)
            out.append(E.PackedConv(conv.weight.unsqueeze(2), s, b, dtype=self.compute_dtype, device=dev))
            self._refresh.fold(bn, conv.bias, s, b)
            self._refresh.pack(out[-1], conv.weight)
        return out

    def packed(self):
        sig = (params_signature(self), self.compute_dtype)
        if self._packed is not None and self._packed_sig != sig and sig[1] == self._packed_sig[1] and E.same_storage(sig[0], self._packed_sig[0]):
            self._refresh.run(self.outc.conv.weight.device)        # updated in place (the other phase's optimizer step): two launches
            self._packed_sig = sig
        if self._packed is None or self._packed_sig != sig:
            dev = self.outc.conv.weight.device
            E.require_cuda(self.outc.conv.weight, "UNet")
            self._refresh = E.PackedRefresh()
            P = {"inc": self._pack_dc(self.inc, dev)}
            for i in (1, 2, 3, 4):
                P["down%d" % i] = self._pack_dc(getattr(self, "down%d" % i).maxpool_conv[1], dev)
                P["up%d" % i] = self._pack_dc(getattr(self, "up%d" % i).conv, dev)
            w = self.outc.conv.weight
            P["outc"] = E.PackedConv(w.unsqueeze(2), torch.ones(w.shape[0]), self.outc.conv.bias, dtype=self.compute_dtype, device=dev)
            self._refresh.pack(P["outc"], w)
            self._refresh.bias(P["outc"], self.outc.conv.bias)
            self._packed, self._packed_sig = P, sig
        return self._packed

    def forward(self, x: torch.Tensor, taps=None) -> torch.Tensor:
        if self.training:      # batch-statistics BatchNorm2d + a tape for loss.backward() (train_anonymizer.py:73,80,92)
            from . import autograd
            return autograd.unet_forward(self, x)
        if x.requires_grad and torch.is_grad_enabled():
            raise NotImplementedError("UNet in eval() mode with a gradient w.r.t. its input: not built (no caller in the reference)")
        E.require_cuda(x, "UNet")
        if x.dim() != 4 or x.shape[1] != self.n_channels:
            raise ValueError("expected (N,%d,H,W), got %s" % (self.n_channels, tuple(x.shape)))
        P = self.packed()
        n, _, H, W = x.shape
        tdt = E.DTYPES[self.compute_dtype][0]
        a = E.clip_to_act(x.unsqueeze(2), cpad=8, dtype=self.compute_dtype)     # (N,1,H,W,8)
        pad = (0, 1, 1)
        # encoder: level i output (channels enc[i]) goes to the left slice of decoder concat buffer cat[i]
        enc_c = (64, 128, 256, 512)
        cats, skips = [], []
        h, w = H, W
        cur = a
        for lvl in range(4):
            pcs = P["inc"] if lvl == 0 else P["down%d" % lvl]
            if lvl > 0:
                cur = E.maxpool(cur, (1, 2, 2), (1, 2, 2))
                h, w = h // 2, w // 2
            cat = E.Act.empty(n, 1, h, w, 2 * enc_c[lvl], tdt, x.device)
            mid = pcs[0](cur, pads=pad)
            skip = pcs[1](mid, pads=pad, out=cat.slice(0, enc_c[lvl]))
            cats.append(cat)
            skips.append(skip)
            cur = skip
        cur = E.maxpool(cur, (1, 2, 2), (1, 2, 2))
        cur = P["down4"][1](P["down4"][0](cur, pads=pad), pads=pad)             # (N,1,H/16,W/16,512)
        for i, lvl in zip((1, 2, 3, 4), (3, 2, 1, 0)):
            cat = cats[lvl]
            _, _, sh_, sw_ = cat.dims
            _, _, ch, cw = cur.dims
            dy, dx = sh_ - 2 * ch, sw_ - 2 * cw                                  # unet_parts.py:58-62
            E.upsample2x_into(cur, cat.slice(enc_c[lvl], enc_c[lvl]), dy // 2, dx // 2)
            pcs = P["up%d" % i]
            cur = pcs[1](pcs[0](cat, pads=pad), pads=pad)
            if taps is not None:
                taps["up%d" % i] = cur
        y = P["outc"](cur, relu=False, sigmoid=True)                             # 1x1 conv + bias + sigmoid
        return E.act_to_nchw(y, self.n_classes).squeeze(2)
