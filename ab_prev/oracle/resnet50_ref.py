"""TEST INFRASTRUCTURE ONLY (CPU oracle; imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg,
never by the product path).

The privacy branch fb = nn.Sequential(torchvision resnet50 with fc = Identity, MLP) of
aux_code/model_loaders.py:124-153 as a functional torch program over a state_dict with the reference's keys
(`0.conv1.weight`, `0.layer1.0.downsample.1.running_mean`, `1.fc2.bias`, ...).

PARITY UNPINNED for the ResNet-50 trunk: torchvision==0.15.2 (pip_requirements.txt:78) is a third-party dependency
that is neither under /root/reference nor installed here; its published `resnet50` (v1.5: stride on the 3x3 conv of a
bottleneck; stem 7x7/2 pad 3; MaxPool2d(3, 2, padding=1); stages [3,4,6,3]; AdaptiveAvgPool2d(1); BN eps 1e-5,
momentum 0.1) is restated. The MLP (model_loaders.py:126-139) is reference source.
"""
import torch
import torch.nn.functional as F

STAGES = ((3, 1), (4, 2), (6, 2), (3, 2))


def _bn(x, sd, p, train, momentum=0.1, eps=1e-5):
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"], train, momentum, eps)


def trunk(x, sd, train=False, prefix="0."):
    """train=True updates the running statistics in `sd` in place, like the module would."""
    p = prefix
    x = F.relu(_bn(F.conv2d(x, sd[p + "conv1.weight"], None, 2, 3), sd, p + "bn1.", train))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, (blocks, stride) in enumerate(STAGES, 1):
        for i in range(blocks):
            q = "%slayer%d.%d." % (p, li, i)
            s = stride if i == 0 else 1
            h = F.relu(_bn(F.conv2d(x, sd[q + "conv1.weight"]), sd, q + "bn1.", train))
            h = F.relu(_bn(F.conv2d(h, sd[q + "conv2.weight"], None, s, 1), sd, q + "bn2.", train))
            h = _bn(F.conv2d(h, sd[q + "conv3.weight"]), sd, q + "bn3.", train)
            if q + "downsample.0.weight" in sd:
                x = _bn(F.conv2d(x, sd[q + "downsample.0.weight"], None, s), sd, q + "downsample.1.", train)
            x = F.relu(h + x)
    return F.adaptive_avg_pool2d(x, 1).flatten(1)


def forward(x, sd, train=False):
    """fb_model(x): (N,3,H,W) -> (N,128) unit-norm embedding."""
    f = trunk(x, sd, train)
    h = F.relu(F.linear(f, sd["1.fc1.weight"], sd["1.fc1.bias"]))
    return F.normalize(F.linear(h, sd["1.fc2.weight"], sd["1.fc2.bias"]), p=2, dim=1)
