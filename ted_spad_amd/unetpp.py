"""The reference's DEFAULT anonymizer `fa` (arch='unet++', aux_code/model_loaders.py:17-30) on MI355X:

    UnetPlusPlus(encoder_name='resnet18', encoder_depth=4, encoder_weights='imagenet', decoder_channels=(256, 128, 64, 32),
                 decoder_attention_type=None, decoder_use_batchnorm=True, in_channels=3, classes=3, activation=None)

`segmentation_models_pytorch==0.3.3` (pip_requirements.txt:65) is a third-party dependency that is NOT under /root/reference and not
installed in this image: the architecture is restated from its published source (base/model.py SegmentationModel.forward,
encoders/resnet.py ResNetEncoder, decoders/unetplusplus/decoder.py UnetPlusPlusDecoder / DecoderBlock, base/heads.py
SegmentationHead) -- PARITY UNPINNED against smp itself; the arithmetic is pinned against oracle/unetpp_ref.py (torch.nn.functional).
The `state_dict` key names are smp's (`encoder.layer1.0.conv1.weight`, `decoder.blocks.x_0_1.conv1.1.running_var`,
`segmentation_head.0.bias`, ... incl. the unused `encoder.layer4.*` that ResNetEncoder keeps at depth 4), so reference
`fa_model_state_dict` checkpoints load with strict=True. `encoder_weights='imagenet'` is a download in smp; there is no network in
this build: the encoder starts from torchvision's random init and says so.

forward((N,3,H,W)) -> (N,3,H,W), H and W multiples of 16 (smp's check_input_shape), NO output activation. In train() the forward
goes through the autograd bridge (autograd.unetpp_forward -> train_nets.UNetPPTrainer: batch-statistics BatchNorm, tape, backward).

Launch plan: every conv3x3 + BN + ReLU is one fused implicit-GEMM launch (2-D = kt 1); BasicBlock tails fuse bn2 + residual + ReLU;
every tensor that is concatenated is written by its producer straight into its channel slice of the consumer's concat buffer, the
nearest x2 upsample writes the leading slice; tensors the dense skip pathway concatenates into TWO blocks (f1, f2, x_2_2) are
copied once per extra use (tedspad_copy_channels).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import _lib, engine as E
from ._lib import check
from .engine import Act, _stream_ptr
from .params import BNParams, ConvParams, params_signature

DECODER_CHANNELS = (256, 128, 64, 32)
ENC_CHANNELS = (3, 64, 64, 128, 256)          # ResNetEncoder.out_channels[:depth + 1] for resnet18, depth 4


class BasicBlock(nn.Module):
    """torchvision.models.resnet.BasicBlock parameter layout (conv1/bn1/conv2/bn2[/downsample.{0,1}])."""

    def __init__(self, inplanes, planes, stride):
        super().__init__()
        self.conv1 = ConvParams(inplanes, planes, (3, 3), init="kaiming_fan_out")
        self.bn1 = BNParams(planes)
        self.conv2 = ConvParams(planes, planes, (3, 3), init="kaiming_fan_out")
        self.bn2 = BNParams(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(ConvParams(inplanes, planes, (1, 1), init="kaiming_fan_out"), BNParams(planes))
        self.stride = stride


class ResNet18Encoder(nn.Module):
    """smp's ResNetEncoder(resnet18): torchvision's ResNet without fc / avgpool; layer4 exists but is not run at depth 4."""

    def __init__(self):
        super().__init__()
        self.conv1 = ConvParams(3, 64, (7, 7), init="kaiming_fan_out")
        self.bn1 = BNParams(64)
        inplanes = 64
        for li, (planes, stride) in enumerate(((64, 1), (128, 2), (256, 2), (512, 2)), 1):
            setattr(self, "layer%d" % li, nn.Sequential(BasicBlock(inplanes, planes, stride), BasicBlock(planes, planes, 1)))
            inplanes = planes


def _conv2d_relu(cin, cout):
    """smp.base.modules.Conv2dReLU with use_batchnorm=True: Sequential(conv (no bias), bn, relu) -> keys 0.weight, 1.*"""
    conv = ConvParams(cin, cout, (3, 3))
    nn.init.kaiming_uniform_(conv.weight, mode="fan_in", nonlinearity="relu")        # smp.base.initialization.initialize_decoder
    return nn.Sequential(conv, BNParams(cout), nn.Identity())


class DecoderBlock(nn.Module):
    def __init__(self, in_channels, skip_channels, out_channels):
        super().__init__()
        self.conv1 = _conv2d_relu(in_channels + skip_channels, out_channels)
        self.conv2 = _conv2d_relu(out_channels, out_channels)
        self.in_channels, self.skip_channels, self.out_channels = in_channels, skip_channels, out_channels


class UnetPlusPlusDecoder(nn.Module):
    """Channel plan of smp's UnetPlusPlusDecoder.__init__ for encoder channels (3,64,64,128,256), n_blocks 4."""

    def __init__(self):
        super().__init__()
        enc = ENC_CHANNELS[1:][::-1]                               # (256, 128, 64, 64)
        in_ch = [enc[0]] + list(DECODER_CHANNELS[:-1])             # [256, 256, 128, 64]
        skip = list(enc[1:]) + [0]                                 # [128, 64, 64, 0]
        out = DECODER_CHANNELS
        blocks = {}
        for layer_idx in range(len(in_ch) - 1):
            for depth_idx in range(layer_idx + 1):
                if depth_idx == 0:
                    i, s, o = in_ch[layer_idx], skip[layer_idx] * (layer_idx + 1), out[layer_idx]
                else:
                    o = skip[layer_idx]
                    s = skip[layer_idx] * (layer_idx + 1 - depth_idx)
                    i = skip[layer_idx - 1]
                blocks["x_%d_%d" % (depth_idx, layer_idx)] = DecoderBlock(i, s, o)
        blocks["x_0_%d" % (len(in_ch) - 1)] = DecoderBlock(in_ch[-1], 0, out[-1])
        self.blocks = nn.ModuleDict(blocks)


class UnetPlusPlus(nn.Module):
    def __init__(self, dtype=E.DEFAULT_DTYPE):
        super().__init__()
        self.encoder = ResNet18Encoder()
        self.decoder = UnetPlusPlusDecoder()
        head = ConvParams(DECODER_CHANNELS[-1], 3, (3, 3), bias=True)
        nn.init.xavier_uniform_(head.weight)                       # smp.base.initialization.initialize_head
        nn.init.constant_(head.bias, 0)
        self.segmentation_head = nn.Sequential(head, nn.Identity(), nn.Identity())
        self.compute_dtype = dtype
        self._packed, self._packed_sig = None, None

    # ---- weights resident in the kernels' layout (BatchNorm folded) ----------------------------------------------------------------
    def packed(self):
        sig = (params_signature(self), self.compute_dtype)
        if self._packed is not None and self._packed_sig != sig and sig[1] == self._packed_sig[1] and E.same_storage(sig[0], self._packed_sig[0]):
            self._refresh.run(self.encoder.conv1.weight.device)    # updated in place (the other phase's optimizer step): two launches
            self._tail_img.copy_(self._pack_tail(self.encoder.conv1.weight.device))     # the fused tail's LDS weight image, rewritten in place
            self._packed_sig = sig
        if self._packed is None or self._packed_sig != sig:
            E.require_cuda(self.encoder.conv1.weight, "UnetPlusPlus")
            dev, dt = self.encoder.conv1.weight.device, self.compute_dtype
            R = self._refresh = E.PackedRefresh()

            def pc(conv, bn, stride=1, pair_w=None):
                s, b = E.fold_bn(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
                c = E.PackedConv(conv.weight.detach().unsqueeze(2), s, b, stride=(1, stride, stride), dtype=dt, device=dev, pair_w=pair_w)
                R.fold(bn, None, s, b)
                R.pack(c, conv.weight)
                return c

            enc = self.encoder
            P = {"stem": pc(enc.conv1, enc.bn1, 2, pair_w=3)}
            for li in (1, 2, 3):                                   # layer4 is not on the path at encoder_depth = 4
                for bi, blk in enumerate(getattr(enc, "layer%d" % li)):
                    p = "layer%d.%d." % (li, bi)
                    P[p + "conv1"] = pc(blk.conv1, blk.bn1, blk.stride)
                    P[p + "conv2"] = pc(blk.conv2, blk.bn2)
                    if blk.downsample is not None:
                        P[p + "down"] = pc(blk.downsample[0], blk.downsample[1], blk.stride)
            for name, blk in self.decoder.blocks.items():
                P[name + ".conv1"] = pc(blk.conv1[0], blk.conv1[1])
                P[name + ".conv2"] = pc(blk.conv2[0], blk.conv2[1])
            head = self.segmentation_head[0]
            P["head"] = E.PackedConv(head.weight.detach().unsqueeze(2), torch.ones(head.weight.shape[0]), head.bias, dtype=dt, device=dev)
            R.pack(P["head"], head.weight)
            R.bias(P["head"], head.bias)
            self._tail_img = self._pack_tail(dev)
            self._packed, self._packed_sig = P, sig
        return self._packed

    def _pack_tail(self, dev):
        """The LDS weight image of tedspad_unetpp_tail_fwd (csrc/conv_upp_tail.hip): x_0_3.conv1 [hc 2][tap 9][co 32][32 ci] | x_0_3.conv2 [tap 9][co 32][32 ci] |
        head [tap 9][co 16 (3 used)][32 ci], the four 16-byte pieces of every 64-byte row stored at piece ^ ((co >> 1) & 3). BatchNorm stays in scale / shift."""
        blk = self.decoder.blocks["x_0_3"]

        def img(w, co_pad, nhc):
            co, ci = w.shape[:2]
            wp = torch.zeros((co_pad, nhc * 32, 3, 3), dtype=torch.float32, device=dev)
            wp[:co, :ci] = w.detach().to(dev, torch.float32)
            t = wp.permute(2, 3, 0, 1).reshape(9, co_pad, nhc, 4, 8).permute(2, 0, 1, 3, 4).contiguous()       # [hc][tap][co][piece][8]
            cidx, ps = torch.arange(co_pad, device=dev), torch.arange(4, device=dev)
            src = ps[None, :] ^ ((cidx[:, None] >> 1) & 3)                                                        # LDS piece slot -> source piece
            return torch.gather(t, 3, src[None, None, :, :, None].expand(nhc, 9, co_pad, 4, 8)).reshape(-1)

        wimg = torch.cat([img(blk.conv1[0].weight, 32, 2), img(blk.conv2[0].weight, 32, 1), img(self.segmentation_head[0].weight, 16, 1)])
        wimg = wimg.to(E.DTYPES[self.compute_dtype][0]).contiguous()
        assert wimg.numel() * 2 == _lib.lib().tedspad_unetpp_tail_wimg_bytes()
        return wimg

    def _tail(self, x02: Act, P) -> torch.Tensor:
        """x_0_3 (interpolate + conv-bn-relu + conv-bn-relu) + segmentation head in one launch: (n, 1, h/2, w/2, 64) -> (n, 3, h, w) fp32."""
        n, _, h2, w2 = x02.dims
        c1, c2, hd = P["x_0_3.conv1"], P["x_0_3.conv2"], P["head"]
        y = torch.empty((n, 3, 2 * h2, 2 * w2), dtype=torch.float32, device=x02.buf.device)
        nc = max(1, min(n, ((1 << 31) - 1) // max(h2 * w2 * x02.ld, 12 * h2 * w2)))
        for n0 in range(0, n, nc):
            n1 = min(n, n0 + nc)
            xs = Act(x02.buf[n0:n1], x02.c, x02.coff)
            check(_lib.lib().tedspad_unetpp_tail_fwd(xs.ptr, xs.ld, y[n0:n1].data_ptr(), n1 - n0, 2 * h2, 2 * w2, self._tail_img.data_ptr(), c1.scale.data_ptr(),
                                                     c1.shift.data_ptr(), c2.scale.data_ptr(), c2.shift.data_ptr(), hd.shift.data_ptr(),
                                                     E.DTYPES[self.compute_dtype][1], _stream_ptr()), "tedspad_unetpp_tail_fwd")
        return y

    # ---- launch sequence ------------------------------------------------------------------------------------------------------------
    @staticmethod
    def _up_into(x: Act, out: Act):
        n, _, h, w = x.dims
        assert out.dims == (n, 1, 2 * h, 2 * w) and out.c == x.c
        check(_lib.lib().tedspad_upsample_nearest2x_fwd(x.ptr, out.ptr, n, h, w, x.c, x.ld, out.ld, _stream_ptr()), "tedspad_upsample_nearest2x_fwd")

    @staticmethod
    def _copy_into(x: Act, out: Act):
        n, _, h, w = x.dims
        assert out.dims == x.dims and out.c == x.c
        check(_lib.lib().tedspad_copy_channels(x.ptr, out.ptr, n * h * w, x.c, x.ld, out.ld, _stream_ptr()), "tedspad_copy_channels")

    def _forward_gathered(self, x: torch.Tensor, P, taps=None) -> torch.Tensor:
        """The eval forward with every decoder block's `interpolate(x, 2, 'nearest')` + `torch.cat([x, *skips])` read IN PLACE by its first conv
        (PackedConv.gather: per-64-channel-chunk sources, half-resolution ones through the x2 index map): no upsampled tensor, no concat buffer, no copies."""
        pad = (0, 1, 1)
        a = E.clip_to_act(x.unsqueeze(2), cpad=4, dtype=self.compute_dtype)
        st = P["stem"]
        f1 = st(a, pads=(0, 3, st.pair_pw), pads_back=(0, 3, st.k[2] - 1 - st.pair_pw))
        cur = E.maxpool(f1, (1, 3, 3), (1, 2, 2), pads=(0, 1, 1))
        feats = {1: f1}
        for li in (1, 2, 3):
            for bi, blk in enumerate(getattr(self.encoder, "layer%d" % li)):
                p = "layer%d.%d." % (li, bi)
                h = P[p + "conv1"](cur, pads=pad)
                res = P[p + "down"](cur, relu=False) if blk.downsample is not None else cur
                cur = P[p + "conv2"](h, pads=pad, residual=res, relu=True)
            feats[li + 1] = cur
        f2, f3, f4 = feats[2], feats[3], feats[4]

        def block(name, *sources):          # sources in smp's torch.cat order: the upsampled input first, then the skips
            return P[name + ".conv2"](P[name + ".conv1"].gather([(sources[0], True)] + [(s, False) for s in sources[1:]], pads=pad), pads=pad)

        x00 = block("x_0_0", f4, f3)
        x11 = block("x_1_1", f3, f2)
        x22 = block("x_2_2", f2, f1)
        x01 = block("x_0_1", x00, x11, f2)
        x12 = block("x_1_2", x11, x22, f1)
        x02 = block("x_0_2", x01, x12, x22, f1)
        if E.UPP_TAIL and taps is None:
            return self._tail(x02, P)
        x03 = block("x_0_3", x02)
        if taps is not None:
            taps.update(f1=f1, f2=f2, f3=f3, f4=f4, x00=x00, x11=x11, x22=x22, x01=x01, x12=x12, x02=x02, x03=x03)
        y = P["head"](x03, pads=pad, relu=False)
        return E.act_to_nchw(y, 3).squeeze(2)

    def forward(self, x: torch.Tensor, taps=None) -> torch.Tensor:
        if self.training:
            # train(): batch-statistics BatchNorm + a tape for loss.backward() (train_anonymizer.py:73-123); under no_grad the same
            # forward runs (running statistics still move, as in torch) and the tape is dropped
            from .autograd import unetpp_forward
            return unetpp_forward(self, x)
        E.require_cuda(x, "UnetPlusPlus")
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("expected (N,3,H,W), got %s" % (tuple(x.shape),))
        n, _, H, W = x.shape
        if H % 16 or W % 16:       # smp.base.model.SegmentationModel.check_input_shape (output stride 2**depth)
            raise RuntimeError("Wrong input shape height=%d, width=%d. Expected image height and width divisible by 16." % (H, W))
        P = self.packed()
        tdt = E.DTYPES[self.compute_dtype][0]
        dev = x.device

        def buf(c, div):
            return Act.empty(n, 1, H // div, W // div, c, tdt, dev)

        pad = (0, 1, 1)
        if E.GATHER_CAT:
            return self._forward_gathered(x, P, taps)
        # concat buffers (leading slice = the upsampled input of the block, then the skips in smp's torch.cat order)
        B00, B11, B22 = buf(384, 8), buf(192, 4), buf(128, 2)      # [up(f4)|f3]  [up(f3)|f2]  [up(f2)|f1]
        B01, B12, B02 = buf(384, 4), buf(192, 2), buf(320, 2)      # [up(x00)|x11|f2]  [up(x11)|x22|f1]  [up(x01)|x12|x22|f1]
        # ---- encoder (smp ResNetEncoder.forward: conv1+bn1+relu | maxpool+layer1 | layer2 | layer3) ----
        a = E.clip_to_act(x.unsqueeze(2), cpad=4, dtype=self.compute_dtype)
        st = P["stem"]
        f1 = st(a, pads=(0, 3, st.pair_pw), pads_back=(0, 3, st.k[2] - 1 - st.pair_pw), out=B22.slice(64, 64))
        cur = E.maxpool(f1, (1, 3, 3), (1, 2, 2), pads=(0, 1, 1))
        outs = {1: B11.slice(128, 64), 2: B00.slice(256, 128), 3: None}
        feats = {1: f1}
        for li in (1, 2, 3):
            layer = getattr(self.encoder, "layer%d" % li)
            for bi, blk in enumerate(layer):
                p = "layer%d.%d." % (li, bi)
                h = P[p + "conv1"](cur, pads=pad)
                res = P[p + "down"](cur, relu=False) if blk.downsample is not None else cur
                last = bi == len(layer) - 1
                cur = P[p + "conv2"](h, pads=pad, residual=res, relu=True, out=outs[li] if last else None)
            feats[li + 1] = cur
        f2, f3, f4 = feats[2], feats[3], feats[4]
        if taps is not None:
            taps.update(f1=f1, f2=f2, f3=f3, f4=f4)

        def block(name, cat: Act, out=None):
            return P[name + ".conv2"](P[name + ".conv1"](cat, pads=pad), pads=pad, out=out)

        # ---- decoder (UnetPlusPlusDecoder.forward, dense skip pathway) ----
        self._up_into(f4, B00.slice(0, 256))
        x00 = block("x_0_0", B00)
        self._up_into(f3, B11.slice(0, 128))
        x11 = block("x_1_1", B11, out=B01.slice(256, 64))
        self._up_into(f2, B22.slice(0, 64))
        x22 = block("x_2_2", B22, out=B12.slice(64, 64))
        self._copy_into(f2, B01.slice(320, 64))
        self._up_into(x00, B01.slice(0, 256))
        x01 = block("x_0_1", B01)
        self._copy_into(f1, B12.slice(128, 64))
        self._up_into(x11, B12.slice(0, 64))
        x12 = block("x_1_2", B12, out=B02.slice(128, 64))
        self._copy_into(Act(B12.buf, 128, B12.coff + 64), B02.slice(192, 128))       # [x22 | f1], contiguous in both buffers
        self._up_into(x01, B02.slice(0, 128))
        x02 = block("x_0_2", B02)
        B03 = buf(64, 1)
        self._up_into(x02, B03)
        x03 = block("x_0_3", B03)
        if taps is not None:
            taps.update(x00=x00, x11=x11, x22=x22, x01=x01, x12=x12, x02=x02, x03=x03)
        y = P["head"](x03, pads=pad, relu=False)                    # SegmentationHead: conv3x3 + bias, no activation
        return E.act_to_nchw(y, 3).squeeze(2)
