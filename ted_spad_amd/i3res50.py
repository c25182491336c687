"""ResNet-50 I3D ("largei3d", 2048-d clip feature) on MI355X.

Mirrors the reference module `I3Res50` (aux_code/models/large_i3d.py:130-263): same
constructor arguments, same `state_dict` key names, `forward(x) -> (logits, feat)` and
`extract_features(x) -> (B, 2048, 1, 1, 1)`. The arithmetic is the HIP implicit-GEMM conv
kernel with the BatchNorm / residual / ReLU epilogue fused (include/tedspad_hip.h); the
network is just the launch sequence below.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import _lib, engine as E
from .params import BNParams, ConvParams, LinearParams, params_signature

# (planes, blocks, spatial stride, temp_conv) -- large_i3d.py:142-145
LAYER_PLAN = ((64, 3, 1, (1, 1, 1)), (128, 4, 2, (1, 0, 1, 0)), (256, 6, 2, (1, 0, 1, 0, 1, 0)), (512, 3, 2, (0, 1, 0)))


class Bottleneck(nn.Module):
    """Parameter layout of large_i3d.py:42-84 (conv1 3x1x1|1x1x1, conv2 1x3x3, conv3 1x1x1)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride, has_down, temp_conv):
        super().__init__()
        self.conv1 = ConvParams(inplanes, planes, (1 + 2 * temp_conv, 1, 1), init="kaiming_fan_out")
        self.bn1 = BNParams(planes)
        self.conv2 = ConvParams(planes, planes, (1, 3, 3), init="kaiming_fan_out")
        self.bn2 = BNParams(planes)
        self.conv3 = ConvParams(planes, planes * 4, (1, 1, 1), init="kaiming_fan_out")
        self.bn3 = BNParams(planes * 4)
        self.downsample = None
        if has_down:
            self.downsample = nn.Sequential(ConvParams(inplanes, planes * 4, (1, 1, 1), init="kaiming_fan_out"),
                                            BNParams(planes * 4))
        self.stride, self.temp_conv = stride, temp_conv


class I3Res50(nn.Module):
    feature_dim = 2048       # width of the clip feature (large_i3d.py:262)

    def __init__(self, num_classes=400, use_nl=False, dtype=E.DEFAULT_DTYPE):
        super().__init__()
        if use_nl:
            raise NotImplementedError("NonLocalBlock is never enabled by the reference (model_loaders.py:262)")
        self.conv1 = ConvParams(3, 64, (5, 7, 7), init="kaiming_fan_out")
        self.bn1 = BNParams(64)
        inplanes = 64
        for li, (planes, blocks, stride, tc) in enumerate(LAYER_PLAN, 1):
            blks = []
            for i in range(blocks):
                blks.append(Bottleneck(inplanes, planes, stride if i == 0 else 1, i == 0, tc[i]))
                inplanes = planes * 4
            setattr(self, "layer%d" % li, nn.Sequential(*blks))
        self.fc = LinearParams(2048, num_classes)
        self.drop_p = 0.5
        self.compute_dtype = dtype
        self._packed = None
        self._packed_sig = None

    # ---- weight packing (BN folded to fp32 scale/shift; 16-bit K-major weights) --------
    def _bn_fold(self, bn: BNParams):
        return E.fold_bn(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)

    def packed(self):
        sig = (params_signature(self), self.compute_dtype)
        if self._packed is None or self._packed_sig != sig:
            dev = self.conv1.weight.device
            E.require_cuda(self.conv1.weight, "I3Res50")
            P = {}
            s, b = self._bn_fold(self.bn1)
            P["stem"] = E.PackedConv(self.conv1.weight, s, b, stride=(2, 2, 2), dtype=self.compute_dtype, device=dev, pair_w=3)
            P["stem_pt"] = E.StemPT(self.conv1.weight, s, b, stride=(2, 2, 2), pads=(2, 3, 3), dtype=self.compute_dtype, device=dev)
            for li in range(1, 5):
                for i, blk in enumerate(getattr(self, "layer%d" % li)):
                    p = "layer%d.%d." % (li, i)
                    s, b = self._bn_fold(blk.bn1)
                    P[p + "conv1"] = E.PackedConv(blk.conv1.weight, s, b, dtype=self.compute_dtype, device=dev)
                    if li >= 2 and blk.temp_conv and E.TPairConv.supported(blk.conv1.weight):      # layers behind maxpool2: T = 2 at 16-frame clips
                        P[p + "conv1_tp"] = E.TPairConv(blk.conv1.weight, s, b, dtype=self.compute_dtype, device=dev)
                    s, b = self._bn_fold(blk.bn2)
                    P[p + "conv2"] = E.PackedConv(blk.conv2.weight, s, b, stride=(1, blk.stride, blk.stride),
                                                  dtype=self.compute_dtype, device=dev)
                    s, b = self._bn_fold(blk.bn3)
                    P[p + "conv3"] = E.PackedConv(blk.conv3.weight, s, b, dtype=self.compute_dtype, device=dev)
                    if blk.downsample is not None:
                        s, b = self._bn_fold(blk.downsample[1])
                        P[p + "down"] = E.PackedConv(blk.downsample[0].weight, s, b, stride=(1, blk.stride, blk.stride),
                                                     dtype=self.compute_dtype, device=dev)
                    if blk.downsample is None and E.BneckFrame.supported(blk.conv1.weight, blk.conv2.weight, blk.conv3.weight):
                        # layer3's plain blocks: the whole bottleneck in one launch, a workgroup per 14 x 14 frame
                        P[p + "frame"] = E.BneckFrame(blk.conv1.weight, *self._bn_fold(blk.bn1), blk.conv2.weight, *self._bn_fold(blk.bn2),
                                                      blk.conv3.weight, *self._bn_fold(blk.bn3), dtype=self.compute_dtype, device=dev)
                    if li in (1, 2) and E.BneckTail.supported(P[p + "conv2"], blk.conv3.weight, blk.downsample[0].weight if blk.downsample is not None else None):
                        s3, b3 = self._bn_fold(blk.bn3)
                        if blk.downsample is not None and blk.stride == 1:
                            sd_, bd_ = self._bn_fold(blk.downsample[1])
                            P[p + "tail"] = E.BneckTail(P[p + "conv2"], blk.conv3.weight, s3, b3, blk.downsample[0].weight, sd_, bd_)
                        elif blk.downsample is None:
                            P[p + "tail"] = E.BneckTail(P[p + "conv2"], blk.conv3.weight, s3, b3)
                    if blk.downsample is not None:
                        if blk.stride == 2:     # conv3 + bn3 and the strided downsample branch as one K-concatenated GEMM
                            s3, b3 = self._bn_fold(blk.bn3)
                            P[p + "dual"] = E.PackedConv.fused_pair(blk.conv3.weight, s3, b3, blk.downsample[0].weight, s, b,
                                                                    dtype=self.compute_dtype, device=dev)
            self._packed, self._packed_sig = P, sig
        return self._packed

    # ---- the launch sequence ---------------------------------------------------------------
    def _trunk(self, x: torch.Tensor, taps=None, records: torch.Tensor = None) -> E.Act:
        """conv1 .. layer4 (large_i3d.py:229-238 == :251-260) on a (B,3,T,H,W) fp32 clip batch -- or, with `records`, on the persistent stem's own 16-bit input
        layout X[n][tp][h][2][w/2][24] (engine.StemPT.layout; preprocess.crop_resize_records writes it straight from uint8 frames)."""
        if self.training:
            raise NotImplementedError("a bare I3Res50 in train() mode has no caller in the reference: training goes through wrapper_i3d "
                                      "(load_ft_model('largei3d'); ted_spad_amd/autograd.py) or train_step.AnonymizerTrainStep")
        P = self.packed()
        if records is not None:
            E.require_cuda(records, "I3Res50")
            if not (E.STEM_PT and E.STEM_POOL) or "stem_pt" not in P:
                raise _lib.TedSpadHipError("I3Res50: the stem-record input needs the persistent stem with the fused pool (TEDSPAD_STEM_PT / TEDSPAD_STEM_POOL are off, "
                                           "or this weight shape has no persistent stem): feed the (B,3,T,H,W) clip instead")
            st = P["stem_pt"]
            if records.dim() != 6 or tuple(records.shape[3:]) != (2, records.shape[4], 24) or records.dtype != st.torch_dtype or not records.is_contiguous():
                raise ValueError("expected contiguous stem records (n, frame pairs, h, 2, w/2, 24) of %s, got %s %s" % (st.torch_dtype, tuple(records.shape), records.dtype))
            if records.shape[1] < 1 or (records.shape[2] + 1) // 2 < 3 or records.shape[4] < 3:
                raise ValueError("stem records need at least one frame pair and frames of 5 x 6 pixels for conv1 + maxpool1, got %s" % (tuple(records.shape),))
            a = st.conv_pool(records)                                    # conv1 + bn1 + ReLU + maxpool1 from the records (the LDS-DMA loader of csrc/conv_stem_pt.hip)
            x = None
        else:
            E.require_cuda(x, "I3Res50")
            if x.dim() != 5 or x.shape[1] != 3:
                raise ValueError("expected (B,3,T,H,W), got %s" % (tuple(x.shape),))
            if x.shape[4] % 2:
                raise ValueError("W must be even")
        if records is not None:
            pass
        elif E.STEM_PT and taps is None and P["stem_pt"].applies(x):
            # conv1 + bn1 + ReLU + maxpool1 on the persistent stem kernel (large_i3d.py:229-232): the 112 x 112 x 8-frame stem tensor
            # is never written
            st = P["stem_pt"]
            if E.STEM_POOL and x.shape[3] >= 5 and x.shape[4] >= 6:
                if E.STEM_CLIP and (st.VARIANT & 4) and st.direct_applies(x):
                    a = st.conv_pool_clip(x)                             # ... straight from the fp32 NCTHW clip: no layout pass either
                else:
                    a = st.conv_pool(st.layout(x))                       # ... and the spatial half: only the pooled tensor is written
            else:
                a = E.maxpool(st(x), (1, 3, 3), (1, 2, 2))               # the spatial half of MaxPool3d((2,3,3), 2)
        else:
            a = E.clip_to_act(x, cpad=4, dtype=self.compute_dtype)       # (B,T,H,W/2, 2px x 4ch)
            a = P["stem"](a, pads=(2, 3, P["stem"].pair_pw), pads_back=(2, 3, 1))   # the same conv in pixel-pair form, K = 5*7*4*8
            if taps is not None:
                taps["stem"] = a
            a = E.maxpool(a, (2, 3, 3), (2, 2, 2))                       # large_i3d.py:138
        if taps is not None:
            taps["maxpool1"] = a
        if self.check_saturation:
            self._sat = E.count_saturated(a, self._sat)
        pooled = False
        for li in range(1, 5):
            if li == 2 and not pooled:
                a = E.maxpool(a, (2, 1, 1), (2, 1, 1))                   # large_i3d.py:139
            layer = getattr(self, "layer%d" % li)
            for i, blk in enumerate(layer):
                p = "layer%d.%d." % (li, i)
                bf = P.get(p + "frame") if taps is None else None
                if bf is not None and bf.applies(a):
                    a = bf(a)                                             # conv1 -> conv2 -> conv3 + residual: only the block input and output touch HBM
                    continue
                tp = P.get(p + "conv1_tp") if taps is None else None
                if tp is not None and tp.applies(a, (blk.temp_conv, 0, 0)):
                    h = tp(a)                                             # two frames: both outputs from ONE K = 2*cin GEMM, no products on zero padding
                else:
                    h = P[p + "conv1"](a, pads=(blk.temp_conv, 0, 0))
                tail = P.get(p + "tail") if (E.BNECK_TAIL and taps is None) else None
                if tail is not None and tail.cmid == 128 and not E.BNECK_TAIL128:
                    tail = None
                fuse_pool = li == 1 and i == len(layer) - 1          # the last layer1 block fuses maxpool2 into its conv3 instead (below)
                if tail is not None and tail.applies(h, (0, 1, 1)) and (not fuse_pool or (E.BNECK_TAIL_POOL and not tail.dual and h.dims[1] % 2 == 0)):
                    # conv2 + bn2 + ReLU + conv3 + bn3 + (residual | downsample branch) + ReLU in one launch: the 64-channel tensor between
                    # the two convolutions is never written (large_i3d.py:69-84); the last block of layer1 pools over frame pairs as
                    # well (maxpool2, large_i3d.py:139)
                    if tail.dual:
                        a = tail(h, pads=(0, 1, 1), x2=a)
                    else:
                        a = tail(h, pads=(0, 1, 1), residual=a, pool_t2=fuse_pool)
                        pooled = pooled or fuse_pool
                    continue
                h = P[p + "conv2"](h, pads=(0, 1, 1))
                if blk.downsample is not None and taps is None and P[p + "conv3"].dual_supported(P[p + "down"], h, a):
                    # layer1.0: conv3 + bn3 and the downsample branch in one launch (the 256-channel downsample tensor
                    # is never written; both branches stay fp32 until the sum)
                    a = P[p + "conv3"].call_dual(h, P[p + "down"], a, relu=True)
                    continue
                if (p + "dual") in P and taps is None and P[p + "dual"].dual_p8_supported(h, a, (blk.stride, blk.stride)):
                    # layer2.0 / 3.0 / 4.0: the same pair as ONE GEMM over [W3*s3 | Wd*sd] on the ping-pong kernel
                    a = P[p + "dual"].call_dual_p8(h, a, (blk.stride, blk.stride), relu=True)
                    continue
                res = P[p + "down"](a, relu=False) if blk.downsample is not None else a
                if li == 1 and i == len(layer) - 1 and taps is None and P[p + "conv3"].pool_t2_supported(h):
                    # the block's tail and maxpool2 (large_i3d.py:139) in one launch: the 256-channel tensor is only
                    # written after the temporal pooling (half the bytes, no separate pool pass)
                    a = P[p + "conv3"].call_pool_t2(h, residual=res, relu=True)
                    pooled = True
                else:
                    a = P[p + "conv3"](h, residual=res, relu=True)       # bn3 + (+= residual) + ReLU fused
            if taps is not None:
                taps["layer%d" % li] = a
            if self.check_saturation:
                self._sat = E.count_saturated(a, self._sat)
        return a

    # ---- f16 head-room, observable -----------------------------------------------------------------------------------
    check_saturation = os.environ.get("TEDSPAD_CHECK_SATURATION", "0") == "1"
    _sat = None

    def saturation_counts(self, reset: bool = True):
        """(elements at +-65504, non-finite elements) seen in the stage outputs (maxpool1, layer1..4) of every forward since the last reset, with
        `check_saturation = True` (or TEDSPAD_CHECK_SATURATION=1): the inference stores saturate instead of overflowing (csrc/common.h), silently -- a released
        checkpoint whose activations outgrow f16 shows up here (one read pass per stage output: off by default). Reads the device counter (a sync)."""
        if self._sat is None:
            return 0, 0
        v = self._sat.cpu().numpy().astype("uint32")
        if reset:
            self._sat.zero_()
        return int(v[0]), int(v[1])

    def extract_features(self, x: torch.Tensor) -> torch.Tensor:
        """large_i3d.py:249-263 -> (B, 2048, 1, 1, 1) fp32."""
        a = self._trunk(x)
        return E.global_avgpool(a).view(x.shape[0], -1, 1, 1, 1)

    def extract_features_records(self, records: torch.Tensor) -> torch.Tensor:
        """`extract_features` of clips handed over as the stem's input records (preprocess.crop_resize_records: uint8 frames -> records in one launch;
        StemPT.layout: an fp32 clip batch -> records) -> (B, 2048, 1, 1, 1) fp32. Bit-identical to extract_features of the fp32 clips the records encode."""
        a = self._trunk(None, records=records)
        return E.global_avgpool(a).view(records.shape[0], -1, 1, 1, 1)

    def forward(self, x: torch.Tensor):
        """large_i3d.py:228-246 -> (logits (B,nc), feat). `feat = x.squeeze()` drops the batch
        dim at B=1 exactly like the reference. Dropout is identity in eval."""
        from . import head
        f = E.global_avgpool(self._trunk(x))
        feat = f.view(x.shape[0], -1, 1, 1, 1).squeeze()
        logits = head.linear(f, self.fc.weight, self.fc.bias)
        return logits, feat
