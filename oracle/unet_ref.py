"""ORACLE (test infrastructure, never shipped as product): CPU restatement of the
reference's 2-D UNet anonymizer `fa` (arch='unet') in plain torch fp32.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Parity is PINNED by tests/test_oracle_golden.py against vectors captured from the
reference itself (tests/golden/make_golden.py).

Follows (reference file:line):
  * DoubleConv   aux_code/models/unet_parts.py:8-25   (conv3x3 pad1 +bias -> BN2d -> ReLU) x2
  * Down         aux_code/models/unet_parts.py:28-39  (MaxPool2d(2) -> DoubleConv)
  * Up           aux_code/models/unet_parts.py:42-68  (bilinear x2 align_corners=True, pad to
                                                       the skip, cat([skip, up]), DoubleConv mid=in/2)
  * OutConv      aux_code/models/unet_parts.py:71-77
  * UNet.forward aux_code/models/unet_model.py:26-37  (sigmoid output)
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5


def _id(t, kind):
    return t


def _bn2d(x, sd, p, train, q=_id):
    if train:
        mean = x.mean((0, 2, 3), keepdim=True)
        var = x.var((0, 2, 3), unbiased=False, keepdim=True)
        from .i3res50_ref import update_running
        update_running(sd, p, mean, var, x.numel() // x.shape[1])       # only when the state dict carries `_track_running`
        # q(., "z"): the precision tests round / adopt the conv output the normalisation READS (the batch statistics come from the unrounded one, as the
        # device's come from its fp32 accumulators); the default hook is the identity
        return (q(x, "z") - mean) / torch.sqrt(var + BN_EPS) * sd[p + "weight"].view(1, -1, 1, 1) + sd[p + "bias"].view(1, -1, 1, 1)
    inv = sd[p + "weight"] / torch.sqrt(sd[p + "running_var"] + BN_EPS)
    sh = sd[p + "bias"] - sd[p + "running_mean"] * inv
    return x * inv.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)


def double_conv(x, sd, p, q=_id, train=False):
    for c, b in (("0.", "1."), ("3.", "4.")):
        x = F.conv2d(q(x, "act"), q(sd[p + c + "weight"], "w"), sd[p + c + "bias"], padding=1)
        x = q(F.relu(_bn2d(x, sd, p + b, train, q)), "act")
    return x


def up(x1, skip, sd, p, q=_id, train=False):
    x1 = F.interpolate(x1, scale_factor=2, mode="bilinear", align_corners=True)
    dy, dx = skip.shape[2] - x1.shape[2], skip.shape[3] - x1.shape[3]
    x1 = F.pad(x1, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
    return double_conv(torch.cat([skip, x1], dim=1), sd, p + "conv.double_conv.", q, train)


def forward(x, sd, q=_id, train=False, taps=None, checkpoint=False):
    """x: (N,3,H,W) fp32 in [0,1] -> (N,3,H,W) in (0,1).
    checkpoint: every level (DoubleConv / Up) under torch.utils.checkpoint -- autograd keeps the level inputs only and recomputes a level's inside in its backward, so
    that the full cfg3 batch (384 pseudo-images of 112 x 112) fits the host memory; same arithmetic, same gradients (not with `_track_running` state dicts: a
    recomputed train-mode forward would update the running statistics twice)."""
    if checkpoint:
        from torch.utils.checkpoint import checkpoint as _ck
        assert not sd.get("_track_running", False)
        dc = lambda t, p: _ck(lambda t_: double_conv(t_, sd, p, q, train), t, use_reentrant=False)
        upf = lambda a, b, p: _ck(lambda a_, b_: up(a_, b_, sd, p, q, train), a, b, use_reentrant=False)
    else:
        dc = lambda t, p: double_conv(t, sd, p, q, train)
        upf = lambda a, b, p: up(a, b, sd, p, q, train)
    x1 = dc(x.requires_grad_() if (checkpoint and not x.requires_grad and x.is_leaf) else x, "inc.double_conv.")
    feats = [x1]
    h = x1
    for i in (1, 2, 3, 4):
        h = dc(F.max_pool2d(h, 2), "down%d.maxpool_conv.1.double_conv." % i)
        feats.append(h)
    for i, skip in zip((1, 2, 3, 4), (feats[3], feats[2], feats[1], feats[0])):
        h = upf(h, skip, "up%d." % i)
        if taps is not None:
            taps["up%d" % i] = h
    logits = F.conv2d(q(h, "act"), q(sd["outc.conv.weight"], "w"), sd["outc.conv.bias"])
    return q(torch.sigmoid(logits), "out")
