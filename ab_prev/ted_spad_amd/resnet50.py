"""The privacy branch fb on MI355X: ResNet-50 (+ the SSL projection MLP) with the parameter names of
`torchvision.models.resnet50` so reference `fb_model_state_dict` checkpoints load with strict=True
(aux_code/model_loaders.py:94-165: `load_fb_model`, `load_privacy_ssl`, `build_resnet_predictor`).

torchvision (==0.15.2, pip_requirements.txt:78) is a third-party dependency that is NOT under /root/reference and not
installed in this image: the architecture is restated from its published definition (ResNet v1.5: 7x7/2 stem, 3x3/2
max-pool pad 1, bottleneck stages [3,4,6,3] with the stride on the 3x3 conv, global average pool, fc) -- PARITY
UNPINNED against torchvision itself; the arithmetic is pinned against oracle/resnet50_ref.py (torch.nn.functional).

An image is a clip with T = 1: every conv runs on the same implicit-GEMM kernels as I3Res50 (kt = 1).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import engine as E
from . import head
from .params import BNParams, ConvParams, LinearParams, params_signature

STAGES = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))


class Bottleneck2d(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride, has_down):
        super().__init__()
        self.conv1 = ConvParams(inplanes, planes, (1, 1), init="kaiming_fan_out")
        self.bn1 = BNParams(planes)
        self.conv2 = ConvParams(planes, planes, (3, 3), init="kaiming_fan_out")
        self.bn2 = BNParams(planes)
        self.conv3 = ConvParams(planes, planes * 4, (1, 1), init="kaiming_fan_out")
        self.bn3 = BNParams(planes * 4)
        self.downsample = None
        if has_down:
            self.downsample = nn.Sequential(ConvParams(inplanes, planes * 4, (1, 1), init="kaiming_fan_out"), BNParams(planes * 4))
        self.stride = stride


class _Identity(nn.Module):
    """`resnet_model.fc = nn.Identity()` (model_loaders.py:144): no parameters."""

    def forward(self, x):
        return x


class ResNet50(nn.Module):
    """`fc`: None -> Identity (SSL trunk), int -> Linear(2048, n) (build_resnet_predictor, model_loaders.py:156-165)."""

    def __init__(self, num_classes=1000, dtype=E.DEFAULT_DTYPE):
        super().__init__()
        self.conv1 = ConvParams(3, 64, (7, 7), init="kaiming_fan_out")
        self.bn1 = BNParams(64)
        inplanes = 64
        for li, (planes, blocks, stride) in enumerate(STAGES, 1):
            blks = []
            for i in range(blocks):
                blks.append(Bottleneck2d(inplanes, planes, stride if i == 0 else 1, i == 0))
                inplanes = planes * 4
            setattr(self, "layer%d" % li, nn.Sequential(*blks))
        self.fc = LinearParams(2048, num_classes) if num_classes else _Identity()
        self.compute_dtype = dtype
        self._packed, self._packed_sig = None, None

    def packed(self):
        sig = (params_signature(self), self.compute_dtype)
        if self._packed is None or self._packed_sig != sig:
            E.require_cuda(self.conv1.weight, "ResNet50")
            dev, dt = self.conv1.weight.device, self.compute_dtype

            def pc(conv, bn, stride=1, pair_w=None):
                s, b = E.fold_bn(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
                return E.PackedConv(conv.weight.detach().unsqueeze(2), s, b, stride=(1, stride, stride), dtype=dt, device=dev, pair_w=pair_w)

            P = {"stem": pc(self.conv1, self.bn1, 2, pair_w=3)}
            for li in range(1, 5):
                for i, blk in enumerate(getattr(self, "layer%d" % li)):
                    p = "layer%d.%d." % (li, i)
                    P[p + "conv1"] = pc(blk.conv1, blk.bn1)
                    P[p + "conv2"] = pc(blk.conv2, blk.bn2, blk.stride)
                    P[p + "conv3"] = pc(blk.conv3, blk.bn3)
                    if blk.downsample is not None:
                        P[p + "down"] = pc(blk.downsample[0], blk.downsample[1], blk.stride)
            self._packed, self._packed_sig = P, sig
        return self._packed

    def features(self, x: torch.Tensor) -> torch.Tensor:
        """(N,3,H,W) fp32 -> (N,2048) fp32: conv1 .. avgpool + flatten (eval-mode BN folded)."""
        if self.training:
            raise NotImplementedError("a bare ResNet50 in train() mode: training goes through the Sequential(ResNet50, MLP) that "
                                      "load_fb_model(ssl=True) returns (ted_spad_amd/autograd.py) or train_step.AnonymizerTrainStep")
        E.require_cuda(x, "ResNet50")
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[3] % 2:
            raise ValueError("expected (N,3,H,W) with even W, got %s" % (tuple(x.shape),))
        P = self.packed()
        a = E.clip_to_act(x.unsqueeze(2), cpad=4, dtype=self.compute_dtype)
        a = P["stem"](a, pads=(0, 3, P["stem"].pair_pw), pads_back=(0, 3, P["stem"].k[2] - 1 - P["stem"].pair_pw))
        a = E.maxpool(a, (1, 3, 3), (1, 2, 2), pads=(0, 1, 1))          # -inf padding == skipped taps
        for li in range(1, 5):
            for i, blk in enumerate(getattr(self, "layer%d" % li)):
                p = "layer%d.%d." % (li, i)
                h = P[p + "conv1"](a)
                h = P[p + "conv2"](h, pads=(0, 1, 1))
                res = P[p + "down"](a, relu=False) if blk.downsample is not None else a
                a = P[p + "conv3"](h, residual=res, relu=True)
        return E.global_avgpool(a)

    def forward(self, x):
        f = self.features(x)
        if isinstance(self.fc, _Identity):
            return f
        return head.linear(f, self.fc.weight, self.fc.bias)


class PrivacyMLP(nn.Module):
    """The `MLP` of load_privacy_ssl (model_loaders.py:126-139): relu(fc1 2048->2048) -> normalize(fc2 2048->128)."""

    def __init__(self, final_embedding_size=128, use_normalization=True):
        super().__init__()
        self.final_embedding_size = final_embedding_size
        self.use_normalization = use_normalization
        self.fc1 = LinearParams(2048, 2048, bias=True)
        self.fc2 = LinearParams(2048, final_embedding_size, bias=True)

    def forward(self, x):
        h = head.linear(x, self.fc1.weight, self.fc1.bias, relu=True)
        return head.l2_normalize(head.linear(h, self.fc2.weight, self.fc2.bias))


class PrivacySSL(nn.Sequential):
    """nn.Sequential(resnet50 with fc = Identity, MLP) of model_loaders.py:124-153 (keys `0.*`, `1.fc{1,2}.*`). In train() mode, or in
    eval() mode with an input that requires grad (phase 1: the NT-Xent gradient flows through the frozen fb into fa,
    train_anonymizer.py:75-84), the pair runs as ONE autograd node (ted_spad_amd/autograd.py)."""

    def forward(self, x):
        if self.training or (x.requires_grad and torch.is_grad_enabled()):
            from . import autograd
            return autograd.fb_forward(self, x)
        return super().forward(x)


def load_privacy_ssl(dtype=E.DEFAULT_DTYPE):
    """model_loaders.py:124-153: nn.Sequential(resnet50 with fc = Identity, MLP); keys `0.*`, `1.fc{1,2}.*`."""
    return PrivacySSL(ResNet50(num_classes=0, dtype=dtype), PrivacyMLP())


def build_resnet_predictor(num_classes=7, pretrained=True, dtype=E.DEFAULT_DTYPE):
    """model_loaders.py:156-165. The ImageNet weights (`ResNet50_Weights.DEFAULT`) are a download; there is no network
    in this build, so `pretrained=True` must be satisfied by loading a checkpoint afterwards."""
    return ResNet50(num_classes=num_classes, dtype=dtype)
