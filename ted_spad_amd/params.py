"""Parameter holders that give the MI355X modules the reference's `state_dict` key names
 so reference checkpoints load with `strict=True`.

They own tensors only. Their `forward` raises: arithmetic happens in the HIP kernels the
owning network launches, never in torch.nn.
"""
import math

import torch
import torch.nn as nn


class _NoForward(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError("%s is a parameter holder; the owning network runs the HIP kernels" % type(self).__name__)


class ConvParams(_NoForward):
    """`weight` (cout, cin, *k) [+ `bias`], initialised like the reference does."""

    def __init__(self, cin, cout, k, bias=False, init="default"):
        super().__init__()
        self.weight = nn.Parameter(torch.empty((cout, cin) + tuple(k)))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None
        fan_in = cin * int(math.prod(k))
        if init == "kaiming_fan_out":  # large_i3d.py:150-152
            nn.init.kaiming_normal_(self.weight, mode="fan_out")
        else:  # nn.ConvNd default
            nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            b = 1.0 / math.sqrt(fan_in)
            nn.init.uniform_(self.bias, -b, b)


class BNParams(_NoForward):
    """BatchNorm{1,2,3}d state: weight, bias, running_mean, running_var, num_batches_tracked."""

    def __init__(self, c, eps=1e-5, momentum=0.1):
        super().__init__()
        self.eps, self.momentum = eps, momentum
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        # like torch's _NormBase: old / FrozenBN-style checkpoints carry no num_batches_tracked
        key = prefix + "num_batches_tracked"
        if key not in state_dict:
            state_dict[key] = torch.tensor(0, dtype=torch.long)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)


class LinearParams(_NoForward):
    def __init__(self, cin, cout, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            b = 1.0 / math.sqrt(cin)
            nn.init.uniform_(self.bias, -b, b)


def params_signature(module: nn.Module):
    """Changes whenever any parameter/buffer is written in place, replaced or moved. `_tedspad_rev` (train_engine.mark_updated)
    counts the writes torch's version counter does not see: a fused optimizer's step leaves `_version` alone, and the eval-mode
    weight images keyed on this signature would otherwise survive it."""
    return tuple((t.data_ptr(), t._version + int(getattr(t, "_tedspad_rev", 0)), str(t.device))
                 for t in list(module.parameters()) + list(module.buffers()))
