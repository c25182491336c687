"""ORACLE (test infrastructure, never shipped as product): CPU restatement of the
reference's ResNet-50 I3D ("largei3d") forward, plain torch fp32 functional ops.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Parity is PINNED: tests/test_oracle_golden.py checks this restatement against golden
vectors captured by importing the reference itself (tests/golden/make_golden.py).

Follows (reference file:line):
  * I3Res50.__init__/_make_layer  aux_code/models/large_i3d.py:130-180  (layer plan)
  * Bottleneck.forward            aux_code/models/large_i3d.py:61-84
  * I3Res50.forward               aux_code/models/large_i3d.py:228-246
  * I3Res50.extract_features      aux_code/models/large_i3d.py:249-263
  * mlp.forward                   aux_code/model_loaders.py:250-254
  * wrapper_i3d.forward           aux_code/model_loaders.py:265-268

Everything is a function of a `state_dict` with the reference's key names
(`conv1.weight`, `layer1.0.bn1.running_var`, ...), so the same weights drive the
reference, this oracle and the HIP path.

`q` is an optional rounding hook `q(tensor, kind)`, kind in {"w", "act", "res"}, used by the
precision tests (tests/test_oracle_golden.py::test_gradient_sensitivity_to_f16_activations,
tests/test_hip_ops.py bottleneck cases) to round where the device rounds; the default is the
identity, i.e. the exact fp32 path.
"""
import torch
import torch.nn.functional as F

# (planes, blocks, spatial stride of block 0, temp_conv pattern)  large_i3d.py:142-145
LAYER_PLAN = (
    ("layer1", 64, 3, 1, (1, 1, 1)),
    ("layer2", 128, 4, 2, (1, 0, 1, 0)),
    ("layer3", 256, 6, 2, (1, 0, 1, 0, 1, 0)),
    ("layer4", 512, 3, 2, (0, 1, 0)),
)
BN_EPS = 1e-5  # nn.BatchNorm3d default, large_i3d.py:48


def _id(t, kind):
    return t


def _bn_eval(x, sd, p, eps=BN_EPS):
    # y = (x - mean) / sqrt(var + eps) * gamma + beta, per channel (dim 1)
    g = sd[p + "weight"] if (p + "weight") in sd else sd[p + "scale"]
    inv = g / torch.sqrt(sd[p + "running_var"] + eps)
    sh = sd[p + "bias"] - sd[p + "running_mean"] * inv
    shape = [1, -1] + [1] * (x.dim() - 2)
    return x * inv.view(shape) + sh.view(shape)


def update_running(sd, p, mean, var, count):
    """nn.BatchNorm's train-mode side effect (momentum 0.1, UNBIASED variance into running_var, num_batches_tracked += 1), applied in place to the state dict
    when it carries `_track_running` = True: multi-iteration trajectories (three updates per ft step, one per fa step) need it, single-step
    parity does not."""
    if not sd.get("_track_running"):
        return
    with torch.no_grad():
        m = 0.1
        sd[p + "running_mean"].mul_(1 - m).add_(m * mean.detach().flatten())
        sd[p + "running_var"].mul_(1 - m).add_(m * var.detach().flatten() * (count / max(count - 1, 1)))
        if p + "num_batches_tracked" in sd:
            sd[p + "num_batches_tracked"] += 1


def _bn_train(x, sd, p, eps=BN_EPS):
    dims = [0] + list(range(2, x.dim()))
    mean = x.mean(dims, keepdim=True)
    var = x.var(dims, unbiased=False, keepdim=True)
    update_running(sd, p, mean, var, x.numel() // x.shape[1])
    shape = [1, -1] + [1] * (x.dim() - 2)
    return (x - mean) / torch.sqrt(var + eps) * sd[p + "weight"].view(shape) + sd[p + "bias"].view(shape)


def device_rounding(dtype=torch.float16):
    """(q, bn_train) that round where the HIP training path rounds (ted_spad_amd/train_engine.conv_bn_act_train and its backward), for the matched-rounding parity
    test of the training gradients: tests/test_hip_train_golden.py::test_phase2_at_cfg3_shape_vs_matched_rounding_oracle.
      forward : weights, the clip, every pre-BatchNorm conv output z and every stored activation y are 16-bit values (straight-through: the fp32 master keeps the
                gradient path); the batch statistics come from the fp32 accumulators, i.e. from the UNROUNDED z, the normalisation reads the rounded one;
      backward: the gradient arriving at every stored activation (d y: the sum over its consumers, rounded once) and at every z (d z) is a 16-bit value;
                parameter gradients are fp32 sums of products of those 16-bit values (the device accumulates them in fp32).
    The head (fc, mlp, both BatchNorm1d) and the losses are exact fp32 on the device: untouched here."""
    def r(t):
        return t.to(dtype).float()

    def q(t, kind):
        if kind == "w":
            return t + (r(t) - t).detach() if t.requires_grad else r(t)
        if getattr(t, "_q16", False):
            return t
        y = t + (r(t) - t).detach() if t.requires_grad else r(t)
        if y.requires_grad:
            y.register_hook(r)
        y._q16 = True
        return y

    def bn_train(x, sd, p, eps=BN_EPS):
        dims = [0] + list(range(2, x.dim()))
        mean = x.mean(dims, keepdim=True)
        var = x.var(dims, unbiased=False, keepdim=True)
        shape = [1, -1] + [1] * (x.dim() - 2)
        return (q(x, "z") - mean) / torch.sqrt(var + eps) * sd[p + "weight"].view(shape) + sd[p + "bias"].view(shape)
    return q, bn_train


def bottleneck(x, sd, p, stride, temp_conv, has_down, q=_id, bn=_bn_eval):
    """large_i3d.py:61-84. `x` is the (possibly higher-precision) residual stream."""
    xin = q(x, "act")
    out = F.conv3d(xin, q(sd[p + "conv1.weight"], "w"), padding=(temp_conv, 0, 0))
    out = q(F.relu(bn(out, sd, p + "bn1.")), "act")
    out = F.conv3d(out, q(sd[p + "conv2.weight"], "w"), stride=(1, stride, stride), padding=(0, 1, 1))
    out = q(F.relu(bn(out, sd, p + "bn2.")), "act")
    out = bn(F.conv3d(out, q(sd[p + "conv3.weight"], "w")), sd, p + "bn3.")
    if has_down:
        res = F.conv3d(xin, q(sd[p + "downsample.0.weight"], "w"), stride=(1, stride, stride))
        res = q(bn(res, sd, p + "downsample.1."), "act")
    else:
        res = x
    return q(F.relu(out + res), "res")


def trunk(x, sd, q=_id, bn=_bn_eval, taps=None):
    """conv1 .. layer4 (large_i3d.py:228-238 == :251-260). x: (B,3,T,H,W) fp32."""
    x = F.conv3d(q(x, "act"), q(sd["conv1.weight"], "w"), stride=(2, 2, 2), padding=(2, 3, 3))
    x = q(F.relu(bn(x, sd, "bn1.")), "act")
    if taps is not None:
        taps["stem"] = x
    x = F.max_pool3d(x, kernel_size=(2, 3, 3), stride=(2, 2, 2))
    if taps is not None:
        taps["maxpool1"] = x
    for name, planes, blocks, stride, tc in LAYER_PLAN:
        if name == "layer2":
            x = F.max_pool3d(x, kernel_size=(2, 1, 1), stride=(2, 1, 1))
        for i in range(blocks):
            p = "%s.%d." % (name, i)
            x = bottleneck(x, sd, p, stride if i == 0 else 1, tc[i], i == 0, q=q, bn=bn)
        if taps is not None:
            taps[name] = x
    return x


def extract_features(x, sd, q=_id, taps=None):
    """I3Res50.extract_features, large_i3d.py:249-263  ->  (B, 2048, 1, 1, 1)."""
    x = trunk(x, sd, q=q, taps=taps)
    return x.mean(dim=(2, 3, 4), keepdim=True)


def forward(x, sd, train=False, frozen_bn=False, q=_id, bn_train=None):
    """I3Res50.forward, large_i3d.py:228-246 -> (logits (B,nc), feat = avgpool.squeeze()).
    frozen_bn: the trunk's BatchNorm3d layers replaced by FrozenBN (large_i3d.py:8-38, `freeze_bn` :30-38): running statistics.
    Dropout(0.5) before fc is stochastic in train mode; the oracle omits it (p -> 0),
    SURVEY.md Q13. `feat` is taken BEFORE dropout, so it is unaffected."""
    x = trunk(x, sd, q=q, bn=(bn_train or _bn_train) if (train and not frozen_bn) else _bn_eval)
    x = x.mean(dim=(2, 3, 4), keepdim=True)
    feat = x.squeeze()
    logits = F.linear(x.flatten(1), sd["fc.weight"], sd["fc.bias"])
    return logits, feat


def mlp(feat, sd, p="mlp.", train=False):
    """model_loaders.py:250-254 (the internal autocast is a no-op on the fp32 CPU path)."""
    bn = _bn_train if train else _bn_eval
    h = F.relu(bn(F.linear(feat, sd[p + "fc1.weight"], sd[p + "fc1.bias"]), sd, p + "bn1."))
    h = bn(F.linear(h, sd[p + "fc2.weight"]), sd, p + "bn2.")
    return F.normalize(h, p=2, dim=1)


def wrapper_forward(x, sd, train=False, frozen_bn=False, q=_id, bn_train=None):
    """wrapper_i3d.forward, model_loaders.py:265-268; sd keys prefixed `i3d.` / `mlp.`."""
    i3d = {k[4:]: v for k, v in sd.items() if k.startswith("i3d.")}
    if sd.get("_track_running"):
        i3d["_track_running"] = True
    pred, feat = forward(x, i3d, train=train, frozen_bn=frozen_bn, q=q, bn_train=bn_train)
    return pred, mlp(feat, sd, "mlp.", train=train)
