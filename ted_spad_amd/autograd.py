"""torch.autograd bridges: the factory's modules in `.train()` (or in `.eval()` with a gradient flowing through them)
inside the reference's OWN training loop --

    fa_model.train(); ft_model.eval(); fb_model.eval()
    output1 = [fb_model(fa_model(v)) for v in inputs_vispr]; loss_fb = NTXentLoss(...)(output1[0], output1[1])
    anon_input = fa_model(inputs_video).reshape(...); output, feat1 = ft_model(inputs1); ...
    loss_fa.backward(); optimizer_fa.step()                                   (train_anonymizer.py:73-123)
    fb_model.train(); ft_model.train(); ...; loss_fb.backward(); loss_ft.backward()      (train_anonymizer.py:137-193)

Each network is ONE autograd node: its forward runs the explicit launch sequence of `train_nets` (all arithmetic in
libtedspad_hip.so) and keeps the tape in `ctx`; its backward runs the hand-written backward sequence and hands the
parameter gradients back to autograd, which accumulates them into `.grad` exactly like for torch.nn modules (several
backward passes into the same parameters -- the three clips of a step -- add up). `train_step.AnonymizerTrainStep`
remains the fast path (no autograd bookkeeping, one flush of the packed weight gradients per step).

Modes, as the reference's modules behave under `train()` / `eval()`:
    UNet / UnetPlusPlus train: batch-statistics BatchNorm2d, running stats updated, parameter gradients
    wrapper_i3d         train: the same + dropout before fc; `freeze_bn(ft)` (train_anonymized_action.py:39-40) -> 'frozen'
                        eval with an input that requires grad (phase 1): BN folded, gradient w.r.t. the INPUT only
                        (the reference also accumulates never-used weight gradients there)
    fb (ResNet-50+MLP)  train / eval-with-input-gradient likewise
"""
from __future__ import annotations

import weakref

import torch

from . import engine as E
from . import train_engine as TE

_LIVE = [0]      # tapes alive (autograd nodes whose backward has not run / been freed)


class _Tape:
    """Carried by ctx: the arena generation check and the live-tape count (the zero arena of a training iteration is
    recycled when the first forward of the NEXT iteration starts, i.e. when no tape of the previous one is alive)."""

    def __init__(self, device):
        if _LIVE[0] == 0:
            TE.ARENA.reset(device)
        _LIVE[0] += 1
        self.gen = TE.ARENA.gen
        weakref.finalize(self, _Tape._dead)

    @staticmethod
    def _dead():
        _LIVE[0] = max(0, _LIVE[0] - 1)

    def check(self):
        if self.gen != TE.ARENA.gen:
            raise RuntimeError("backward through a ted_spad_amd module after its training arena was recycled: the tape of a "
                               "previous iteration is no longer valid (call backward before the next iteration's forward)")


def _capture_grads(params, fn):
    """Run `fn` (a hand-written backward that ACCUMULATES into p.grad) and return what it added, leaving p.grad as it was:
    autograd does the accumulation."""
    saved = [p.grad for p in params]
    for p in params:
        p.grad = None
    try:
        out = fn()
        TE.flush_deferred()
        new = [p.grad for p in params]
    finally:
        for p, g in zip(params, saved):
            p.grad = g
    return out, new


def _up(g):
    """A gradient entering a network: scaled up (TE.GRAD_SCALE, see there)."""
    return None if g is None else g.contiguous().float() * TE.GRAD_SCALE


def _down(gs):
    """Gradients leaving a network (fresh tensors): the scale divided out again."""
    if gs is None or TE.GRAD_SCALE == 1.0:
        return gs
    inv = 1.0 / TE.GRAD_SCALE
    if torch.is_tensor(gs):
        return gs * inv
    return [None if g is None else g.mul_(inv) for g in gs]


def _trainer(module, cls):
    tr = module.__dict__.get("_hip_trainer")
    if tr is None:
        E.apply_env_determinism()
        tr = cls(module)
        module.__dict__["_hip_trainer"] = tr
    return tr


# ---- UNet ------------------------------------------------------------------------------------------------------------

class _UNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, trainer, x, *params):
        ctx.tape_guard = _Tape(x.device)
        y, tape = trainer.forward(x)
        ctx.trainer, ctx.tape = trainer, tape
        ctx.set_materialize_grads(False)
        return y

    @staticmethod
    def backward(ctx, dy):
        tr = ctx.trainer
        params = list(tr.m.parameters())
        if dy is None:
            return (None, None) + (None,) * len(params)
        ctx.tape_guard.check()

        def run():
            tr.backward(ctx.tape, _up(dy))
            tr.flush_grads()

        _, grads = _capture_grads(params, run)
        ctx.tape = None
        return (None, None) + tuple(_down(grads))


def unet_forward(module, x):
    from .train_nets import UNetTrainer
    if x.requires_grad and torch.is_grad_enabled():
        raise NotImplementedError("UNet: gradient w.r.t. the INPUT frames is not built (the reference never needs it: fa's input "
                                  "is data, train_anonymizer.py:80,92)")
    tr = _trainer(module, UNetTrainer)
    return _UNetFn.apply(tr, x, *module.parameters())


def unetpp_forward(module, x):
    """UnetPlusPlus (the default `fa`) in train(): the same autograd node, UNetPPTrainer underneath."""
    from .train_nets import UNetPPTrainer
    if x.requires_grad and torch.is_grad_enabled():
        raise NotImplementedError("UnetPlusPlus: gradient w.r.t. the INPUT frames is not built (the reference never needs it: fa's input "
                                  "is data, train_anonymizer.py:80,92)")
    tr = _trainer(module, UNetPPTrainer)
    return _UNetFn.apply(tr, x, *module.parameters())


# ---- wrapper_i3d -------------------------------------------------------------------------------------------------------

class _I3DFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, trainer, mode, drop_mask, x, *params):
        ctx.tape_guard = _Tape(x.device)
        pred, feat, tape = trainer.forward(x, mode, drop_mask=drop_mask)
        ctx.trainer, ctx.tape, ctx.mode = trainer, tape, mode
        ctx.set_materialize_grads(False)
        return pred, feat

    @staticmethod
    def backward(ctx, dpred, dfeat):
        tr, mode = ctx.trainer, ctx.mode
        params = list(tr.m.parameters()) if mode != "eval" else []
        none = (None, None, None)
        if dpred is None and dfeat is None:
            return none + (None,) + (None,) * len(params)
        ctx.tape_guard.check()
        if mode == "eval":
            dx = tr.backward(ctx.tape, _up(dpred), _up(dfeat))
            ctx.tape = None
            return none + (_down(dx),)

        def run():
            tr.backward(ctx.tape, _up(dpred), _up(dfeat))
            tr.flush_grads()

        _, grads = _capture_grads(params, run)
        ctx.tape = None
        return none + (None,) + tuple(_down(grads))


def wrapper_forward(module, x, drop_mask=None):
    """wrapper_i3d.forward with a tape: returns (pred, feat) as autograd-tracked tensors."""
    from .train_nets import I3DTrainer
    tr = _trainer(module, I3DTrainer)
    if module.training:
        mode = "frozen" if getattr(module, "frozen_bn", False) else "train"
        return _I3DFn.apply(tr, mode, drop_mask, x, *module.parameters())
    return _I3DFn.apply(tr, "eval", None, x)


# ---- fb: ResNet-50 + projection MLP -----------------------------------------------------------------------------------------

class _FBFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, trainer, mode, x, *params):
        ctx.tape_guard = _Tape(x.device)
        emb, tape = trainer.forward(x, mode)
        ctx.trainer, ctx.tape, ctx.mode = trainer, tape, mode
        ctx.set_materialize_grads(False)
        return emb

    @staticmethod
    def backward(ctx, demb):
        tr, mode = ctx.trainer, ctx.mode
        params = list(tr.m.parameters()) if mode == "train" else []
        if demb is None:
            return (None, None, None) + (None,) * len(params)
        ctx.tape_guard.check()
        if mode == "eval":
            dx = tr.backward(ctx.tape, _up(demb))
            ctx.tape = None
            return (None, None, _down(dx))

        def run():
            tr.backward(ctx.tape, _up(demb))
            tr.flush_grads()

        _, grads = _capture_grads(params, run)
        ctx.tape = None
        return (None, None, None) + tuple(_down(grads))


def fb_forward(module, x):
    """nn.Sequential(ResNet50(fc = Identity), MLP) with a tape."""
    from .train_nets import FBTrainer
    tr = _trainer(module, FBTrainer)
    if module.training:
        return _FBFn.apply(tr, "train", x, *module.parameters())
    return _FBFn.apply(tr, "eval", x)


def freeze_bn(model):
    """Counterpart of `freeze_bn(ft_model)` (aux_code/models/large_i3d.py:23-38, used by train_anonymized_action.py:39-40): the
    trunk's BatchNorm3d layers use their running statistics and their gamma / beta get no gradient while the module is in
    train() mode; dropout and the mlp head keep following the train flag. Here a flag on the wrapper (the parameters keep
    their names, so checkpoints stay loadable)."""
    target = model if hasattr(model, "i3d") else None
    if target is None:
        raise NotImplementedError("freeze_bn: expected the wrapper_i3d that load_ft_model('largei3d') returns")
    target.frozen_bn = True
    return model
