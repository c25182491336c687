"""ORACLE (test infrastructure, never shipped as product): CPU restatement of the
reference's Inception-v1 I3D ("i3d", 1024-d) forward in plain torch fp32.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Parity is PINNED by tests/test_oracle_golden.py against vectors captured from the
reference itself (tests/golden/make_golden.py).

Follows (reference file:line):
  * TF-"SAME" padding rule      aux_code/models/i3d.py:82-86 (Unit3D.compute_pad), :15-19
  * Unit3D.forward              aux_code/models/i3d.py:89-120   (pad zeros -> conv -> BN eps 1e-3 -> ReLU)
  * MaxPool3dSamePadding        aux_code/models/i3d.py:21-45    (ZERO padding, then max-pool)
  * InceptionModule.forward     aux_code/models/i3d.py:144-149
  * layer plan                  aux_code/models/i3d.py:220-289
  * forward / extract_features  aux_code/models/i3d.py:324-340
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-3  # i3d.py:80

# (name, kind, args): "unit" (cin, cout, k, s) | "pool" (k, s) | "mixed" (cin, [6 widths])
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


def same_pad(size, k, s):
    """(front, back) zero padding of one dim -- i3d.py:82-86 + :101-106."""
    total = max(k - s, 0) if size % s == 0 else max(k - (size % s), 0)
    return total // 2, total - total // 2


def _pad_same(x, k, s):
    pads = []
    for d in (2, 1, 0):  # F.pad order: W, H, T
        f, b = same_pad(x.shape[2 + d], k[d], s[d])
        pads += [f, b]
    return F.pad(x, pads)


def _id(t, kind):
    return t


def unit3d(x, sd, p, k, s, q=_id, bn=True, relu=True):
    x = _pad_same(q(x, "act"), k, s)
    x = F.conv3d(x, q(sd[p + "conv3d.weight"], "w"), sd.get(p + "conv3d.bias"), stride=s)
    if bn:
        inv = sd[p + "bn.weight"] / torch.sqrt(sd[p + "bn.running_var"] + BN_EPS)
        sh = sd[p + "bn.bias"] - sd[p + "bn.running_mean"] * inv
        x = x * inv.view(1, -1, 1, 1, 1) + sh.view(1, -1, 1, 1, 1)
    return q(F.relu(x), "act") if relu else x


def maxpool_same(x, k, s):
    return F.max_pool3d(_pad_same(x, k, s), kernel_size=k, stride=s)


def mixed(x, sd, p, q=_id):
    one, three = (1, 1, 1), (3, 3, 3)
    b0 = unit3d(x, sd, p + "b0.", one, one, q)
    b1 = unit3d(unit3d(x, sd, p + "b1a.", one, one, q), sd, p + "b1b.", three, one, q)
    b2 = unit3d(unit3d(x, sd, p + "b2a.", one, one, q), sd, p + "b2b.", three, one, q)
    b3 = unit3d(maxpool_same(x, three, one), sd, p + "b3b.", one, one, q)
    return torch.cat([b0, b1, b2, b3], dim=1)


def trunk(x, sd, q=_id, taps=None):
    for name, kind, a in PLAN:
        if kind == "unit":
            x = unit3d(x, sd, name + ".", a[2], a[3], q)
        elif kind == "pool":
            x = maxpool_same(x, a[0], a[1])
        else:
            x = mixed(x, sd, name + ".", q)
        if taps is not None:
            taps[name] = x
    return x


def extract_features(x, sd, q=_id, taps=None):
    """i3d.py:336-340: AvgPool3d([2,7,7], stride 1) of the Mixed_5c map -> (B,1024,1,1,1)
    for a 16x224x224 clip; raises for maps smaller than (2,7,7) exactly as the reference."""
    x = trunk(x, sd, q=q, taps=taps)
    return F.avg_pool3d(x, kernel_size=(2, 7, 7), stride=(1, 1, 1))


def forward(x, sd):
    """i3d.py:324-333 (eval: dropout is identity): adaptive avg -> 1x1x1 logits conv with
    bias -> squeeze -> (B, num_classes). Returns ONE tensor."""
    x = trunk(x, sd)
    x = x.mean(dim=(2, 3, 4), keepdim=True)
    x = unit3d(x, sd, "logits.", (1, 1, 1), (1, 1, 1), bn=False, relu=False)
    return x.squeeze(3).squeeze(3).squeeze(2)
