"""Shared helper of the CPU tests: gradients of the fp32 oracle with and without f16 rounding of the forward activations."""
import torch

from oracle import i3res50_ref
from ted_spad_amd.synth import synth_train_video


class _RoundSTE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.half().float()

    @staticmethod
    def backward(ctx, g):
        return g


def _q_half(t, kind):
    return _RoundSTE.apply(t) if kind in ("act", "res") else t.half().float()


def rounded_forward_gradients(sd, B=2, hw=48):
    video = synth_train_video(0, "sens", (B, 16, 3, hw, hw)).permute(0, 2, 1, 3, 4).contiguous()
    labels = torch.tensor([5, 77, 101, 1][:B])

    def run(q):
        p = {k: (v.clone().requires_grad_() if v.is_floating_point() and not k.endswith(("running_mean", "running_var")) else v) for k, v in sd.items()}
        i3 = {k[4:]: v for k, v in p.items() if k.startswith("i3d.")}
        x = i3res50_ref.trunk(video, i3, q=q, bn=i3res50_ref._bn_train)
        pred = torch.nn.functional.linear(x.mean(dim=(2, 3, 4)), i3["fc.weight"], i3["fc.bias"])
        loss = torch.nn.functional.cross_entropy(pred, labels)
        loss.backward()
        return float(loss), {k: v.grad for k, v in p.items() if v.requires_grad and v.grad is not None}

    l0, g0 = run(i3res50_ref._id)
    l1, g1 = run(_q_half)
    errs, cos = [], []
    for k in g0:
        a, b = g1[k].flatten().double(), g0[k].flatten().double()
        if float(b.norm()) > 1e-4:
            errs.append(float((a - b).norm() / b.norm()))
            cos.append(float(a @ b / (a.norm() * b.norm())))
    return l0, l1, errs, cos
