"""Drop-in for the reference's model factory `aux_code/model_loaders.py`: same function
names, arguments, return conventions, print messages and checkpoint-key fallbacks, with
the networks running on MI355X HIP kernels (libtedspad_hip.so).

    load_fa_model(saved_model_file=None, arch='unet++')                       model_loaders.py:17-52
    load_ft_model(arch='r3d', saved_model_file=None, num_classes=400,
                  kin_pretrained=False)                                       model_loaders.py:56-90
    mlp, wrapper_i3d                                                          model_loaders.py:235-268

In scope: arch 'largei3d' and 'i3d' for ft, 'unet' for fa (the architectures whose source is part of the reference), 'unet++' for
fa (segmentation_models_pytorch's UnetPlusPlus, the reference's default, restated: unetpp.py; inference and training) and 'r50' for fb
(torchvision's ResNet-50, restated: resnet50.py). 'r3d_18' and 'mvitv2' are third-party torchvision video models that are out of
scope: they raise NotImplementedError rather than silently falling back.
"""
from __future__ import annotations

import os
from collections import OrderedDict

import torch
import torch.nn as nn

from . import engine as E
from . import head
from .i3res50 import I3Res50
from .params import BNParams, LinearParams


class mlp(nn.Module):
    """2048 -> 512 -> 128 projection, BN1d after each Linear, L2-normalised output
    (model_loaders.py:235-254). State-dict keys: fc1.{weight,bias}, bn1.*, bn2.*, fc2.weight."""

    def __init__(self, final_embedding_size=128, use_normalization=True):
        super().__init__()
        self.final_embedding_size = final_embedding_size
        self.use_normalization = use_normalization
        self.fc1 = LinearParams(2048, 512, bias=True)
        self.bn1 = BNParams(512)
        self.bn2 = BNParams(128)
        self.fc2 = LinearParams(512, final_embedding_size, bias=False)

    def forward(self, x):
        if self.training:
            raise NotImplementedError("train-mode mlp runs inside wrapper_i3d.forward (one autograd node for trunk + head, "
                                      "ted_spad_amd/autograd.py) or AnonymizerTrainStep; a stand-alone train-mode call has no caller in the reference")
        if x.dim() != 2:
            # the reference's BatchNorm1d raises here too when I3Res50.forward squeezed B=1 away
            raise ValueError("mlp expects (B, 2048) with B >= 2, got %s" % (tuple(x.shape),))
        h = head.linear(x, self.fc1.weight, self.fc1.bias, bn=self.bn1, relu=True)
        h = head.linear(h, self.fc2.weight, None, bn=self.bn2, relu=False)
        return head.l2_normalize(h)


class wrapper_i3d(nn.Module):
    """model_loaders.py:258-268: returns (pred, mlp(feature)); keys `i3d.*`, `mlp.*`."""

    def __init__(self, num_classes=102, dtype=E.DEFAULT_DTYPE):
        super().__init__()
        self.i3d = I3Res50(num_classes=num_classes, use_nl=False, dtype=dtype)
        self.mlp = mlp()

    def forward(self, x):
        if self.training or (x.requires_grad and torch.is_grad_enabled()):
            # train(): batch-statistics BN, dropout, parameter gradients (train_anonymizer.py:139,166-179); eval() with an input
            # that requires grad: the frozen ft of phase 1, gradient w.r.t. the clip only (:74,99-112,122)
            from . import autograd
            return autograd.wrapper_forward(self, x)
        pred, feature = self.i3d(x)
        feature = self.mlp(feature)
        return pred, feature


# where the Kinetics checkpoints live: the reference reads them relative to its scripts' working directory (model_loaders.py:178,192)
SAVED_MODELS_DIR = os.environ.get("TEDSPAD_SAVED_MODELS", os.path.join("..", "saved_models"))


def _kinetics_classifier(make, weights_file, target, new_head, num_classes, pretrained):
    """The rule both Kinetics-pretrained classifiers follow (model_loaders.py:171-196): when the checkpoint is to be loaded the network is built with
    Kinetics' 400 classes so that it loads strictly, and only then gets the head for `num_classes`; without it the network is built for `num_classes` directly."""
    model = make(400 if pretrained else num_classes)
    if pretrained:
        target(model).load_state_dict(torch.load(os.path.join(SAVED_MODELS_DIR, weights_file)), strict=True)
        if num_classes != 400:
            new_head(model, num_classes)
    return model


def build_i3d_classifier(num_classes=400, pretrained=True):
    """model_loaders.py:171-182: InceptionI3d, `rgb_imagenet.pt`, head replaced through `replace_logits`."""
    from .inception_i3d import InceptionI3d
    return _kinetics_classifier(lambda n: InceptionI3d(num_classes=n, dropout_keep_prob=0.5), "rgb_imagenet.pt", lambda m: m,
                                lambda m, n: m.replace_logits(n), num_classes, pretrained)


def build_largei3d_classifier(num_classes=400, pretrained=True):
    """model_loaders.py:185-196: wrapper_i3d around I3Res50, `i3d_r50_kinetics.pth` loaded into `.i3d`, a fresh `fc` for the new classes."""
    def new_fc(m, n):
        m.i3d.fc = LinearParams(512 * 4, n)
    return _kinetics_classifier(lambda n: wrapper_i3d(num_classes=n), "i3d_r50_kinetics.pth", lambda m: m.i3d, new_fc, num_classes, pretrained)


def _strip_module(sd):
    return OrderedDict((k[7:], v) for k, v in sd.items())  # remove 'module.' (DataParallel checkpoints)


def load_fa_model(saved_model_file=None, arch="unet++"):
    if arch == "unet++":
        # smp's UnetPlusPlus(resnet18, depth 4, (256,128,64,32), batch-norm decoder, 3 classes, no activation) restated from its
        # published source (unetpp.py); `encoder_weights="imagenet"` is a download there -- offline the encoder is randomly initialised
        from .unetpp import UnetPlusPlus
        fa_model = UnetPlusPlus()
    elif arch == "unet":
        from .unet import UNet
        fa_model = UNet(n_channels=3, n_classes=3)
    else:
        print(f"Architecture {arch} invalid for fa_model. Try 'unet' or 'unet++'")
        return None
    if saved_model_file:
        saved_dict = torch.load(saved_model_file)
        try:
            fa_model.load_state_dict(saved_dict["fa_model_state_dict"], strict=True)
        except Exception:
            fa_model.load_state_dict(_strip_module(saved_dict["fa_model_state_dict"]), strict=True)
        print(f"fa_model loaded from {saved_model_file} successfully!")
    else:
        print("fa_model freshly initialized!")
    return fa_model


def load_ft_model(arch="r3d", saved_model_file=None, num_classes=400, kin_pretrained=False):
    if arch == "i3d":
        ft_model = build_i3d_classifier(num_classes=num_classes, pretrained=kin_pretrained)
    elif arch == "largei3d":
        ft_model = build_largei3d_classifier(num_classes=num_classes, pretrained=kin_pretrained)
    elif arch in ("mvitv2", "r3d_18"):
        raise NotImplementedError("arch '%s' is a torchvision model (third-party): out of scope." % arch)
    else:
        print(f"Architecture {arch} invalid for ft_model. Try 'i3d', 'largei3d', 'mvitv2', or 'r3d_18'.")
        return
    if saved_model_file:
        saved_dict = torch.load(saved_model_file)
        try:
            ft_model.load_state_dict(saved_dict["ft_model_state_dict"], strict=True)
        except Exception:
            try:
                new_state_dict = OrderedDict((k.replace("scale", "weight"), v)  # FrozenBN-style Kinetics ckpts (:80)
                                             for k, v in saved_dict["ft_model_state_dict"].items())
                ft_model.load_state_dict(new_state_dict, strict=True)
            except Exception:
                ft_model.i3d.load_state_dict(saved_dict["ft_model_state_dict"], strict=True)
        print(f"ft_model loaded from {saved_model_file} successfully!")
    else:
        print(f"ft_model freshly initialized! Pretrained: {kin_pretrained}")
    return ft_model


def load_fb_model(arch="r50", saved_model_file=None, num_pa=7, ssl=False, pretrained=True):
    """model_loaders.py:94-120: ResNet-50 privacy branch (ssl=True: + projection MLP, what train_anonymizer.py:338 uses).
    `pretrained=True` asks torchvision for its ImageNet download in the reference; offline, the weights come from
    `saved_model_file` (or stay randomly initialised, and the message says so)."""
    from .resnet50 import build_resnet_predictor, load_privacy_ssl
    if arch == "r50":
        fb_model = load_privacy_ssl() if ssl else build_resnet_predictor(num_classes=num_pa, pretrained=pretrained)
    else:
        print(f"Architecture {arch} invalid for fb_model. Try 'r50'")
        return
    if saved_model_file:
        saved_dict = torch.load(saved_model_file, map_location="cpu")
        try:
            fb_model.load_state_dict(saved_dict["fb_model_state_dict"], strict=True)
        except Exception:
            new_state_dict = OrderedDict((k[7:], v) for k, v in saved_dict["fb_model_state_dict"].items())   # 'module.' (:110-113)
            fb_model.load_state_dict(new_state_dict, strict=True)
        print(f"fb_model loaded from {saved_model_file} successfully!")
    else:
        # the reference prints `Pretrained: {pretrained}` after torchvision downloaded the ImageNet weights; offline
        # nothing was downloaded, so the message states what actually happened
        print("fb_model freshly initialized! Pretrained: False")
    return fb_model
