"""Inception-v1 I3D ("i3d", 1024-d clip feature) on MI355X.

Mirrors the reference module `InceptionI3d` (aux_code/models/i3d.py:152-340): same
constructor arguments, `state_dict` key names (`Mixed_3b.b1b.conv3d.weight`, ...; `logits`
registered FIRST), `forward(x) -> logits` (a single tensor, Q6),
`extract_features(x) -> (B,1024,1,1,1)` and `replace_logits`.

Every Unit3D (TF-SAME zero pad -> conv -> BN eps 1e-3 -> ReLU, i3d.py:89-120) is ONE launch
of the fused implicit-GEMM kernel with asymmetric front/back padding by predication; the four
branches of an Inception block write straight into their channel slice of the block output
(no torch.cat copy, i3d.py:149).
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import engine as E
from . import head
from .params import BNParams, ConvParams, params_signature

# (name, kind, args) -- i3d.py:220-289
PLAN = (
    ("Conv3d_1a_7x7", "unit", (3, 64, (7, 7, 7), (2, 2, 2))),
    ("MaxPool3d_2a_3x3", "pool", ((1, 3, 3), (1, 2, 2))),
    ("Conv3d_2b_1x1", "unit", (64, 64, (1, 1, 1), (1, 1, 1))),
    ("Conv3d_2c_3x3", "unit", (64, 192, (3, 3, 3), (1, 1, 1))),
    ("MaxPool3d_3a_3x3", "pool", ((1, 3, 3), (1, 2, 2))),
    ("Mixed_3b", "mixed", (192, (64, 96, 128, 16, 32, 32))),
    ("Mixed_3c", "mixed", (256, (128, 128, 192, 32, 96, 64))),
    ("MaxPool3d_4a_3x3", "pool", ((3, 3, 3), (2, 2, 2))),
    ("Mixed_4b", "mixed", (480, (192, 96, 208, 16, 48, 64))),
    ("Mixed_4c", "mixed", (512, (160, 112, 224, 24, 64, 64))),
    ("Mixed_4d", "mixed", (512, (128, 128, 256, 24, 64, 64))),
    ("Mixed_4e", "mixed", (512, (112, 144, 288, 32, 64, 64))),
    ("Mixed_4f", "mixed", (528, (256, 160, 320, 32, 128, 128))),
    ("MaxPool3d_5a_2x2", "pool", ((2, 2, 2), (2, 2, 2))),
    ("Mixed_5b", "mixed", (832, (256, 160, 320, 32, 128, 128))),
    ("Mixed_5c", "mixed", (832, (384, 192, 384, 48, 128, 128))),
)


FUSE_REDUCE = os.environ.get("TEDSPAD_I3D_FUSE_REDUCE", "1") != "0"   # Mixed_*: the b1a / b2a 1x1x1 convs as one GEMM (0: two launches, A/B)
FUSE_B0 = os.environ.get("TEDSPAD_I3D_FUSE_B0", "1") != "0"           # ... and branch 0's 1x1x1 conv in the same GEMM, writing its concat slice directly (0: its own launch, A/B)


class Unit3D(nn.Module):
    def __init__(self, cin, cout, k=(1, 1, 1), s=(1, 1, 1), use_batch_norm=True, use_bias=False):
        super().__init__()
        self.conv3d = ConvParams(cin, cout, tuple(k), bias=use_bias)
        self.bn = BNParams(cout, eps=1e-3, momentum=0.01) if use_batch_norm else None  # i3d.py:80
        self.k, self.s = tuple(k), tuple(s)


class InceptionModule(nn.Module):
    def __init__(self, cin, oc):
        super().__init__()
        self.b0 = Unit3D(cin, oc[0])
        self.b1a = Unit3D(cin, oc[1])
        self.b1b = Unit3D(oc[1], oc[2], (3, 3, 3))
        self.b2a = Unit3D(cin, oc[3])
        self.b2b = Unit3D(oc[3], oc[4], (3, 3, 3))
        self.b3b = Unit3D(cin, oc[5])
        self.oc = tuple(oc)


class InceptionI3d(nn.Module):
    feature_dim = 1024       # width of the clip feature (i3d.py:336-340)
    VALID_ENDPOINTS = tuple(n for n, _, _ in PLAN) + ("Logits", "Predictions")

    def __init__(self, num_classes=400, spatial_squeeze=True, final_endpoint="Logits", name="inception_i3d",
                 in_channels=3, dropout_keep_prob=0.5, dtype=E.DEFAULT_DTYPE):
        super().__init__()
        if final_endpoint != "Logits" or in_channels != 3:
            raise NotImplementedError("only the full RGB network (final_endpoint='Logits') is used by the reference")
        self._num_classes = num_classes
        self._spatial_squeeze = spatial_squeeze
        self.logits = Unit3D(1024, num_classes, use_batch_norm=False, use_bias=True)  # registered first (Q16)
        for name_, kind, a in PLAN:
            if kind == "unit":
                self.add_module(name_, Unit3D(a[0], a[1], a[2], a[3]))
            elif kind == "mixed":
                self.add_module(name_, InceptionModule(a[0], a[1]))
        self.compute_dtype = dtype
        self._packed = None
        self._packed_sig = None

    def replace_logits(self, num_classes):
        """i3d.py:309-317."""
        self._num_classes = num_classes
        dev = self.logits.conv3d.weight.device
        self.logits = Unit3D(1024, num_classes, use_batch_norm=False, use_bias=True).to(dev)

    # ---- packing ------------------------------------------------------------------------------
    def _pack_unit(self, u: Unit3D, dev, pair_w=None):
        s, b = E.fold_bn(u.bn.weight, u.bn.bias, u.bn.running_mean, u.bn.running_var, u.bn.eps)
        return E.PackedConv(u.conv3d.weight, s, b, stride=u.s, dtype=self.compute_dtype, device=dev, pair_w=pair_w)

    def packed(self):
        sig = (params_signature(self), self.compute_dtype)
        if self._packed is None or self._packed_sig != sig:
            dev = self.logits.conv3d.weight.device
            E.require_cuda(self.logits.conv3d.weight, "InceptionI3d")
            P = {}
            for name_, kind, a in PLAN:
                m = getattr(self, name_, None)
                if kind == "unit":
                    # the 7x7x7 stride-2 stem: TF-SAME on even sizes pads (2,3) -> pixel-pair form with front pad 2
                    P[name_] = self._pack_unit(m, dev, pair_w=2 if a[0] == 3 else None)
                elif kind == "mixed":
                    for b in ("b0", "b1a", "b1b", "b2a", "b2b", "b3b"):
                        P[name_ + "." + b] = self._pack_unit(getattr(m, b), dev)
                    # the two 1x1x1 "reduce" convs in front of the 3x3x3 branches read the same tensor: ONE GEMM over [W_1a ; W_2a] whose output
                    # the two 3x3x3 convs read as channel slices (the module input is read twice instead of three times, one launch less)
                    ua, ub = m.b1a, m.b2a
                    if ua.conv3d.weight.shape[0] % 8 == 0 and tuple(ua.s) == tuple(ub.s) == (1, 1, 1):
                        sa, ba = E.fold_bn(ua.bn.weight, ua.bn.bias, ua.bn.running_mean, ua.bn.running_var, ua.bn.eps)
                        sb, bb = E.fold_bn(ub.bn.weight, ub.bn.bias, ub.bn.running_mean, ub.bn.running_var, ub.bn.eps)
                        P[name_ + ".b12a"] = E.PackedConv(torch.cat([ua.conv3d.weight, ub.conv3d.weight]), torch.cat([sa, sb]), torch.cat([ba, bb]),
                                                          dtype=self.compute_dtype, device=dev)
                        # ... and branch 0 (i3d.py:144-149: all three read the module input x): ONE GEMM over [W_1a ; W_2a ; W_0]. Its output channels are
                        # contiguous in a buffer laid out [t1 | t2 | b0 | b1b | b2b | b3b]: the module's output is the channel slice behind t2 (no copy)
                        s0, b0_ = E.fold_bn(m.b0.bn.weight, m.b0.bn.bias, m.b0.bn.running_mean, m.b0.bn.running_var, m.b0.bn.eps)
                        P[name_ + ".b012a"] = E.PackedConv(torch.cat([ua.conv3d.weight, ub.conv3d.weight, m.b0.conv3d.weight]), torch.cat([sa, sb, s0]),
                                                           torch.cat([ba, bb, b0_]), dtype=self.compute_dtype, device=dev)
            self._packed, self._packed_sig = P, sig
        return self._packed

    # ---- launch sequence ------------------------------------------------------------------------
    @staticmethod
    def _same(dims, k, s):
        pf, pb = zip(*(E.same_pads(d, kk, ss) for d, kk, ss in zip(dims, k, s)))
        return tuple(pf), tuple(pb)

    def _unit(self, pc, x: E.Act, k, s, out=None):
        pf, pb = self._same(x.dims[1:], k, s)
        return pc(x, pads=pf, pads_back=pb, out=out)

    def _trunk(self, x: torch.Tensor, taps=None) -> E.Act:
        if self.training:
            raise NotImplementedError("train-mode InceptionI3d is unusable in the reference too (SURVEY.md Q6); call .eval()")
        E.require_cuda(x, "InceptionI3d")
        if x.dim() != 5 or x.shape[1] != 3:
            raise ValueError("expected (B,3,T,H,W), got %s" % (tuple(x.shape),))
        if x.shape[4] % 2:
            raise ValueError("W must be even")
        P = self.packed()
        a = E.clip_to_act(x, cpad=4, dtype=self.compute_dtype)
        one, three = (1, 1, 1), (3, 3, 3)
        for name_, kind, arg in PLAN:
            if kind == "unit" and arg[0] == 3:
                pc = P[name_]
                (pt, ph, _), (bt, bh, _) = self._same((x.shape[2], x.shape[3], x.shape[4]), arg[2], arg[3])
                a = pc(a, pads=(pt, ph, pc.pair_pw), pads_back=(bt, bh, pc.k[2] - 1 - pc.pair_pw))
            elif kind == "unit":
                a = self._unit(P[name_], a, arg[2], arg[3])
            elif kind == "pool":
                pf, pb = self._same(a.dims[1:], arg[0], arg[1])
                a = E.maxpool(a, arg[0], arg[1], pf, pb, pad_zero=True)           # ZERO pad, then max (i3d.py:41-45)
            else:
                oc = arg[1]
                n, t, h, w = a.dims
                OC = oc[0] + oc[2] + oc[4] + oc[5]
                if (name_ + ".b012a") in P and taps is None and FUSE_REDUCE and FUSE_B0:
                    # [pad | t1 | t2 | b0 | b1b | b2b | b3b]: the reduce tensors in front of the module's output, padded so that the output slice starts on a 128-byte line
                    red = oc[1] + oc[3]
                    off = (red + 63) // 64 * 64
                    big = E.Act.empty(n, t, h, w, off + OC, a.buf.dtype, a.buf.device)
                    self._unit(P[name_ + ".b012a"], a, one, one, out=E.Act(big.buf, red + oc[0], off - red))
                    t1, t2 = E.Act(big.buf, oc[1], off - red), E.Act(big.buf, oc[3], off - red + oc[1])
                    out = E.Act(big.buf, OC, off)
                    self._unit(P[name_ + ".b1b"], t1, three, one, out=out.slice(oc[0], oc[2]))
                    self._unit(P[name_ + ".b2b"], t2, three, one, out=out.slice(oc[0] + oc[2], oc[4]))
                    pf, pb = self._same(a.dims[1:], three, one)
                    t3 = E.maxpool(a, three, one, pf, pb, pad_zero=True)
                    self._unit(P[name_ + ".b3b"], t3, one, one, out=out.slice(oc[0] + oc[2] + oc[4], oc[5]))
                    a = out
                    continue
                out = E.Act.empty(n, t, h, w, OC, a.buf.dtype, a.buf.device)
                self._unit(P[name_ + ".b0"], a, one, one, out=out.slice(0, oc[0]))
                if (name_ + ".b12a") in P and taps is None and FUSE_REDUCE:
                    t12 = self._unit(P[name_ + ".b12a"], a, one, one)
                    t1, t2 = t12.slice(0, oc[1]), t12.slice(oc[1], oc[3])
                else:
                    t1 = self._unit(P[name_ + ".b1a"], a, one, one)
                    t2 = self._unit(P[name_ + ".b2a"], a, one, one)
                self._unit(P[name_ + ".b1b"], t1, three, one, out=out.slice(oc[0], oc[2]))
                self._unit(P[name_ + ".b2b"], t2, three, one, out=out.slice(oc[0] + oc[2], oc[4]))
                pf, pb = self._same(a.dims[1:], three, one)
                t3 = E.maxpool(a, three, one, pf, pb, pad_zero=True)
                self._unit(P[name_ + ".b3b"], t3, one, one, out=out.slice(oc[0] + oc[2] + oc[4], oc[5]))
                a = out
            if taps is not None:
                taps[name_] = a
        return a

    def extract_features(self, x: torch.Tensor) -> torch.Tensor:
        """i3d.py:293-295,336-340: AvgPool3d([2,7,7], stride 1) of Mixed_5c. A 16x224x224 clip gives a (2,7,7) map ->
        (B,1024,1,1,1); larger clips give (B,1024,t-1,h-6,w-6) like the reference; smaller maps raise like it."""
        a = self._trunk(x)
        _, t, h, w = a.dims
        if t < 2 or h < 7 or w < 7:
            raise RuntimeError("AvgPool3d kernel (2,7,7) is larger than the Mixed_5c map (%d,%d,%d)" % (t, h, w))
        if (t, h, w) == (2, 7, 7):
            return E.global_avgpool(a).view(x.shape[0], -1, 1, 1, 1)
        return E.avgpool3d_stride1(a, (2, 7, 7))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """i3d.py:324-333 (eval): adaptive avg-pool -> 1x1x1 logits conv with bias -> (B, num_classes)."""
        f = E.global_avgpool(self._trunk(x))
        w = self.logits.conv3d.weight
        return head.linear(f, w.reshape(w.shape[0], -1), self.logits.conv3d.bias)
