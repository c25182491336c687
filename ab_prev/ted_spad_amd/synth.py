"""Portable, counter-based synthetic data generator (weights, BN statistics, clips).

Neither `torch.manual_seed` nor `numpy.random` streams are guaranteed stable across
versions/devices, so every synthetic tensor in this repo (golden fixtures, tests, bench
inputs) is produced by a stateless hash: element i of tensor `name` under `seed` is

    u24 = splitmix64(key(seed, name) + (i + 1) * GOLDEN) >> 40        # top 24 bits
    x   = u24 / 2**24                                                 # exact in fp32

The same function exists for numpy (host) and torch (any device, int64 wrap-around
arithmetic), and both produce bit-identical fp32 values, so the GPU box can regenerate
exactly the inputs the golden fixtures were made from without shipping them.

There is no reference counterpart: the reference initialises from Kinetics checkpoints
that are not distributable; random-init weights of the same
architecture stand in (BASELINE.json `configs`).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch

_GOLDEN = 0x9E3779B97F4A7C15
_M1 = 0xBF58476D1CE4E5B9
_M2 = 0x94D049BB133111EB
_MASK = (1 << 64) - 1


def _fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & _MASK
    return h


def _key(seed: int, name: str) -> int:
    return (_fnv1a64(name) ^ ((int(seed) * 0xD1342543DE82EF95 + 0x2545F4914F6CDD1D) & _MASK)) & _MASK


def uniform01_np(seed: int, name: str, n: int, offset: int = 0) -> np.ndarray:
    """n fp32 values in [0,1), elements offset..offset+n-1 of stream (seed, name)."""
    key = np.uint64(_key(seed, name))
    i = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = key + i * np.uint64(_GOLDEN)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(_M1)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(_M2)
        z = z ^ (z >> np.uint64(31))
    return ((z >> np.uint64(40)).astype(np.float32)) * np.float32(1.0 / (1 << 24))


def _s64(v: int) -> int:
    v &= _MASK
    return v - (1 << 64) if v >= (1 << 63) else v


def _lsr(x: torch.Tensor, s: int) -> torch.Tensor:
    # logical shift right on int64 (torch's >> is arithmetic)
    return (x >> s) & ((1 << (64 - s)) - 1)


def uniform01_torch(seed: int, name: str, n: int, device="cpu", offset: int = 0) -> torch.Tensor:
    """Bit-identical to `uniform01_np`, computed on `device` with wrapping int64 math."""
    key = _s64(_key(seed, name))
    i = torch.arange(offset + 1, offset + n + 1, dtype=torch.int64, device=device)
    z = i * _s64(_GOLDEN) + key
    z = (z ^ _lsr(z, 30)) * _s64(_M1)
    z = (z ^ _lsr(z, 27)) * _s64(_M2)
    z = z ^ _lsr(z, 31)
    return _lsr(z, 40).to(torch.float32) * (1.0 / (1 << 24))


def synth_tensor(seed: int, name: str, shape, lo=0.0, hi=1.0, device="cpu") -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    if str(device) == "cpu":
        u = torch.from_numpy(uniform01_np(seed, name, n))
    else:
        u = uniform01_torch(seed, name, n, device=device)
    return (u * (hi - lo) + lo).reshape(tuple(shape))


def synth_clips(seed: int, n_clips: int, shape=(3, 16, 224, 224), device="cpu", first: int = 0) -> torch.Tensor:
    """Clips `first .. first+n_clips-1` of the synthetic video `seed`: fp32 uniform [0,1),
    layout (n, 3, T, H, W) as `ft.extract_features` takes it. Clip g is stream
    ("clip", offset g*numel) so any rank can generate exactly its own shard."""
    per = int(np.prod(shape))
    name = "clip/%dx%dx%dx%d" % tuple(shape)
    if str(device) == "cpu":
        u = torch.from_numpy(uniform01_np(seed, name, n_clips * per, offset=first * per))
    else:
        u = uniform01_torch(seed, name, n_clips * per, device=device, offset=first * per)
    return u.reshape((n_clips,) + tuple(shape))


def synth_state_dict(template: "OrderedDict[str, torch.Tensor]", seed: int = 0,
                     residual_gamma: float = 0.5) -> "OrderedDict[str, torch.Tensor]":
    """Fill a state_dict (names + shapes taken from `template`) with portable values.

    * conv / linear weights: uniform(-a, a) with He scaling, a = sqrt(6 / fan_in)
      (variance 2 / fan_in); linear layers use a = sqrt(3 / fan_in).
    * BatchNorm: gamma in [0.5, 1.5], beta in [-0.2, 0.2], running_mean in [-0.1, 0.1],
      running_var in [0.5, 1.5] -- non-identity, so BN folding is really exercised.
      The BN that closes a residual branch (`bn3`) has gamma scaled by `residual_gamma`
      so the 16-block residual stream stays O(1)-O(10), as in a trained network.
    * other biases: uniform(-0.1, 0.1); `num_batches_tracked` = 0.
    """
    bn_prefixes = {k[: -len("running_mean")] for k in template if k.endswith("running_mean")}
    out = OrderedDict()
    for k, t in template.items():
        shape = tuple(t.shape)
        prefix = k[: k.rfind(".") + 1]
        leaf = k[k.rfind(".") + 1:]
        if leaf == "num_batches_tracked":
            out[k] = torch.zeros(shape, dtype=t.dtype)
            continue
        if prefix in bn_prefixes:
            if leaf in ("weight", "scale"):
                v = synth_tensor(seed, k, shape, 0.5, 1.5)
                if prefix.endswith("bn3."):
                    v = v * residual_gamma
            elif leaf == "bias":
                v = synth_tensor(seed, k, shape, -0.2, 0.2)
            elif leaf == "running_mean":
                v = synth_tensor(seed, k, shape, -0.1, 0.1)
            elif leaf == "running_var":
                v = synth_tensor(seed, k, shape, 0.5, 1.5)
            else:
                raise KeyError(k)
        elif leaf == "weight" and len(shape) >= 3:
            fan_in = int(np.prod(shape[1:]))
            a = math.sqrt(6.0 / fan_in)
            v = synth_tensor(seed, k, shape, -a, a)
        elif leaf == "weight" and len(shape) == 2:
            a = math.sqrt(3.0 / shape[1])
            v = synth_tensor(seed, k, shape, -a, a)
        elif leaf == "bias":
            v = synth_tensor(seed, k, shape, -0.1, 0.1)
        else:
            raise KeyError("synth_state_dict: no rule for %s %s" % (k, shape))
        out[k] = v.to(t.dtype)
    return out


def synth_train_video(seed: int, name: str, shape, device="cpu") -> torch.Tensor:
    """(B,48,3,H,W) training batch in [0,1]: i.i.d. noise frames with a per-sample brightness gain
    (b+1)/B. Pure i.i.d. noise makes the pooled features of all samples nearly identical, and a
    train-mode BatchNorm over such a batch amplifies rounding noise instead of signal."""
    b = shape[0]
    gain = (torch.arange(1, b + 1, dtype=torch.float32, device=device) / b).view(b, 1, 1, 1, 1)
    return synth_tensor(seed, name, shape, device=device) * gain
