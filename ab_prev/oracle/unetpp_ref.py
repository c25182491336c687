"""ORACLE (test infrastructure, never shipped as product): CPU restatement, in plain torch fp32, of the reference's default
anonymizer `fa` -- segmentation_models_pytorch's UnetPlusPlus with the arguments of aux_code/model_loaders.py:17-30
(resnet18 encoder, depth 4, decoder channels (256,128,64,32), batch-norm decoder, no attention, 3 classes, no activation).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

PARITY UNPINNED: `segmentation-models-pytorch==0.3.3` (pip_requirements.txt:65) and `torchvision==0.15.2` (:78) are third-party
packages that are neither under /root/reference nor installed in this image, so no golden vector could be captured through them.
The algorithm is restated from their published sources:
  * smp/base/model.py         SegmentationModel.forward: check_input_shape (H, W % 2**depth), encoder -> decoder -> segmentation_head
  * smp/encoders/resnet.py    ResNetEncoder.forward: stages [Identity | conv1+bn1+relu | maxpool+layer1 | layer2 | layer3 | layer4][:depth+1]
  * torchvision/models/resnet.py  BasicBlock: conv3x3(s) bn relu conv3x3 bn (+ downsample(x) = conv1x1(s) bn) relu; conv1 7x7/2 pad 3,
                              MaxPool2d(3, 2, 1)
  * smp/decoders/unetplusplus/decoder.py  DecoderBlock.forward (F.interpolate x2 nearest, cat([x, skip]), Conv2dReLU x2) and
                              UnetPlusPlusDecoder.forward (dense skip pathway x_{depth}_{layer})
  * smp/base/heads.py         SegmentationHead: Conv2d(32, 3, 3, padding=1) + Identity + Identity
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5


def _bn(x, sd, p, train=False):
    if train:      # nn.BatchNorm2d in train(): batch statistics, running statistics updated in place (momentum 0.1)
        return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"], True, 0.1, BN_EPS)
    inv = sd[p + "weight"] / torch.sqrt(sd[p + "running_var"] + BN_EPS)
    sh = sd[p + "bias"] - sd[p + "running_mean"] * inv
    return x * inv.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)


def _basic_block(x, sd, p, stride, train=False):
    out = F.relu(_bn(F.conv2d(x, sd[p + "conv1.weight"], stride=stride, padding=1), sd, p + "bn1.", train))
    out = _bn(F.conv2d(out, sd[p + "conv2.weight"], padding=1), sd, p + "bn2.", train)
    idt = x
    if p + "downsample.0.weight" in sd:
        idt = _bn(F.conv2d(x, sd[p + "downsample.0.weight"], stride=stride), sd, p + "downsample.1.", train)
    return F.relu(out + idt)


def encoder(x, sd, train=False):
    """-> [f0 = x, f1 (64, /2), f2 (64, /4), f3 (128, /8), f4 (256, /16)]"""
    f1 = F.relu(_bn(F.conv2d(x, sd["encoder.conv1.weight"], stride=2, padding=3), sd, "encoder.bn1.", train))
    h = F.max_pool2d(f1, 3, 2, 1)
    feats = [x, f1]
    for li, stride in ((1, 1), (2, 2), (3, 2)):
        for bi in (0, 1):
            h = _basic_block(h, sd, "encoder.layer%d.%d." % (li, bi), stride if bi == 0 else 1, train)
        feats.append(h)
    return feats


def _block(x, skip, sd, name, train=False):
    x = F.interpolate(x, scale_factor=2, mode="nearest")
    if skip is not None:
        x = torch.cat([x, skip], dim=1)
    p = "decoder.blocks.%s." % name
    for c in ("conv1.", "conv2."):
        x = F.relu(_bn(F.conv2d(x, sd[p + c + "0.weight"], padding=1), sd, p + c + "1.", train))
    return x


def decoder(feats, sd, taps=None, train=False):
    features = feats[1:][::-1]           # [f4, f3, f2, f1]
    depth = 3
    dense = {}
    for layer_idx in range(3):
        for depth_idx in range(depth - layer_idx):
            if layer_idx == 0:
                dense["x_%d_%d" % (depth_idx, depth_idx)] = _block(features[depth_idx], features[depth_idx + 1], sd, "x_%d_%d" % (depth_idx, depth_idx), train)
            else:
                dli = depth_idx + layer_idx
                cat = [dense["x_%d_%d" % (idx, dli)] for idx in range(depth_idx + 1, dli + 1)]
                cat = torch.cat(cat + [features[dli + 1]], dim=1)
                dense["x_%d_%d" % (depth_idx, dli)] = _block(dense["x_%d_%d" % (depth_idx, dli - 1)], cat, sd, "x_%d_%d" % (depth_idx, dli), train)
    dense["x_0_3"] = _block(dense["x_0_2"], None, sd, "x_0_3", train)
    if taps is not None:
        taps.update(dense)
    return dense["x_0_3"]


def forward(x, sd, taps=None, train=False):
    """x: (N,3,H,W) fp32, H and W multiples of 16 -> (N,3,H,W) (no activation). train: the module in train() mode (batch-statistics
    BatchNorm; the running statistics in `sd` are updated in place, once per call)."""
    if x.shape[2] % 16 or x.shape[3] % 16:
        raise RuntimeError("Wrong input shape height=%d, width=%d. Expected image height and width divisible by 16." % (x.shape[2], x.shape[3]))
    feats = encoder(x, sd, train)
    if taps is not None:
        taps.update(f1=feats[1], f2=feats[2], f3=feats[3], f4=feats[4])
    h = decoder(feats, sd, taps, train)
    return F.conv2d(h, sd["segmentation_head.0.weight"], sd["segmentation_head.0.bias"], padding=1)
