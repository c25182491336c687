"""ORACLE (test infrastructure, never shipped): CPU restatement of one iteration of the
anonymizer training loss algebra with torch autograd (float32), on the oracle networks.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Pinned by tests/test_oracle_golden.py::test_train_step_* against loss values and gradient norms
captured by running the REFERENCE modules (model_loaders.load_fa_model('unet'),
load_ft_model('largei3d')) through the same lines of train_anonymizer.py (tests/golden/make_golden.py g7).

Follows anonymization_training/train_anonymizer.py:
  feed (Q2)           :57,87-92        split :94       ft x3 :99,111-112
  loss_ft             :107,115-116     loss_fa :119
  fb / NT-Xent term   :80-84 (phase 1: through the frozen fb into fa), :147,153-157,190 (phase 2: updates fb);
                      fb = oracle/resnet50_ref.py (torchvision ResNet-50 restated: that trunk's parity is unpinned)
  phase 2             :137-183
"""
import torch

from . import i3res50_ref, losses_ref, resnet50_ref, unet_ref, unetpp_ref


def _fa(x, sd, train, q=None, checkpoint=False):
    """The anonymizer the state dict belongs to: UNet (arch='unet') or the default smp UnetPlusPlus (model_loaders.py:17-30). q: the UNet's rounding hook."""
    if "encoder.conv1.weight" in sd:
        return unetpp_ref.forward(x, sd, train=train)
    return unet_ref.forward(x, sd, train=train, q=q or unet_ref._id, checkpoint=checkpoint)


def _grad_sd(sd):
    """Leaf copies of the trainable tensors; buffers (running statistics, counters) and the `_track_running` flag are passed through BY REFERENCE, so that a
    tracked train-mode forward updates the caller's state dict in place (i3res50_ref.update_running)."""
    return {k: (v.clone().requires_grad_() if torch.is_tensor(v) and v.is_floating_point() and not k.endswith(("running_mean", "running_var")) else v)
            for k, v in sd.items()}


def _utility(ft_sd, clips, labels, train, tlw=0.1, frozen_bn=False, rounding=None):
    def kw(k):      # rounding: None | (q, bn_train) | a callable clip index -> (q, bn_train)
        if rounding is None:
            return {}
        q, bn_train = rounding(k) if callable(rounding) else rounding
        return dict(q=q, bn_train=bn_train)
    heads = [i3res50_ref.wrapper_forward(c, ft_sd, train=train, frozen_bn=frozen_bn, **kw(k)) for k, c in enumerate(clips)]
    ce = losses_ref.cross_entropy_torch(heads[0][0], labels)
    trip = losses_ref.triplet_torch(heads[0][1], heads[1][1], heads[2][1])
    return ce + tlw * trip, ce, trip


def phase1(video_b48, labels, fa_sd, ft_sd, ft_loss_weight=0.7, tlw=0.1, num_frames=16, vispr=None, fb_sd=None, fb_loss_weight=1.0, fa_rounding=None, ft_rounding=None,
           fb_fn=None, checkpoint=False):
    """Returns (losses dict, grads of fa parameters dict, d(loss)/d(anon)). With `vispr` = [view0, view1] (N,3,H,W)
    and `fb_sd`, the privacy term -fb_loss_weight * NTXent(fb(fa(v0)), fb(fa(v1))) is included (fa in train mode on
    each view separately, fb in eval mode: :73-84); `fb_fn` replaces the ResNet-50 by any callable image -> unit-norm embedding (golden g7's stub). fa_rounding: the UNet's hook q(tensor, kind); ft_rounding: as in `phase2` (per clip) -- the forward-point parity test
    hands both the device's forward tensors."""
    fa = _grad_sd(fa_sd)
    loss_fb = None
    if vispr is not None:
        z = [fb_fn(_fa(x, fa, True)) if fb_fn is not None else resnet50_ref.forward(_fa(x, fa, True), fb_sd, train=False) for x in vispr]
        loss_fb = losses_ref.nt_xent_torch(z[0], z[1], 0.1)
    v = video_b48.permute(0, 2, 1, 3, 4)
    b, c, t, h, w = v.shape
    anon = _fa(v.reshape(-1, c, h, w), fa, True, q=fa_rounding, checkpoint=checkpoint).reshape(b, c, t, h, w)      # checkpoint: the full cfg3 batch on the host
    anon.retain_grad()
    clips = torch.split(anon, [num_frames] * 3, dim=2)
    loss_ft, ce, trip = _utility(ft_sd, clips, labels, train=False, tlw=tlw, rounding=ft_rounding)
    loss_fa = ft_loss_weight * loss_ft
    if loss_fb is not None:
        loss_fa = -fb_loss_weight * loss_fb + loss_fa
    loss_fa.backward()
    grads = {k: p.grad for k, p in fa.items() if torch.is_tensor(p) and p.requires_grad and p.grad is not None}
    return dict(loss_fa=loss_fa.item(), loss_ft=loss_ft.item(), loss_ce=ce.item(), loss_temporal=trip.item(),
                loss_fb=None if loss_fb is None else loss_fb.item()), grads, anon.grad


def phase2_fb(vispr, fa_sd, fb_sd):
    """The fb half of phase 2 (:147,153-157,190): fa eval / no grad on each view, fb in train mode per view, NT-Xent.
    Returns (loss_fb, grads of fb parameters)."""
    fb = _grad_sd(fb_sd)
    with torch.no_grad():
        x = [_fa(v, fa_sd, False) for v in vispr]
    z = [resnet50_ref.forward(xi, fb, train=True) for xi in x]
    loss_fb = losses_ref.nt_xent_torch(z[0], z[1], 0.1)
    loss_fb.backward()
    return loss_fb.item(), {k: p.grad for k, p in fb.items() if torch.is_tensor(p) and p.requires_grad and p.grad is not None}


def phase2(video_b48, labels, fa_sd, ft_sd, tlw=0.1, num_frames=16, ft_rounding=None, anon=None):
    """ft_rounding: `i3res50_ref.device_rounding(dtype)` -- ft's trunk rounds its forward values and activation gradients to 16 bits where the HIP training path
    does (matched-rounding parity); None: the exact fp32 path of the reference. anon: the (B,3,48,H,W) anonymised video to feed ft instead of this
    oracle's own fa output (the matched test hands over the device's, so that ft sees the same input on both sides)."""
    ft = _grad_sd(ft_sd)
    v = video_b48.permute(0, 2, 1, 3, 4)
    b, c, t, h, w = v.shape
    if anon is None:
        with torch.no_grad():
            anon = _fa(v.reshape(-1, c, h, w), fa_sd, False).reshape(b, c, t, h, w)
    clips = torch.split(anon, [num_frames] * 3, dim=2)
    loss_ft, ce, trip = _utility(ft, clips, labels, train=True, tlw=tlw, rounding=ft_rounding)
    loss_ft.backward()
    grads = {k: p.grad for k, p in ft.items() if torch.is_tensor(p) and p.requires_grad and p.grad is not None}
    return dict(loss_ft=loss_ft.item(), loss_ce=ce.item(), loss_temporal=trip.item()), grads


def action_step(video_b48, labels, fa_sd, ft_sd, tlw=0.1, num_frames=16):
    """One iteration of action_training/train_anonymized_action.py:43-94 (`--temporal_loss trip`, cross-entropy loss):
    fa under no_grad (:52-57), ft with its trunk BatchNorm3d layers frozen (`freeze_bn`, :39-40; FrozenBN keeps gamma / beta
    as BUFFERS, large_i3d.py:15-20, so they get no gradient: dropped from the returned dict), head in train mode.
    Parity of this restatement is unpinned (no golden vector was captured through the reference for this script)."""
    ft = _grad_sd(ft_sd)
    v = video_b48.permute(0, 2, 1, 3, 4)
    b, c, t, h, w = v.shape
    with torch.no_grad():
        anon = _fa(v.reshape(-1, c, h, w), fa_sd, False).reshape(b, c, t, h, w)
    clips = torch.split(anon, [num_frames] * 3, dim=2)
    loss, ce, trip = _utility(ft, clips, labels, train=True, tlw=tlw, frozen_bn=True)
    loss.backward()
    frozen = {k for k in ft if k.startswith("i3d.") and (".bn" in k or k.startswith("i3d.bn") or ".downsample.1." in k)}
    grads = {k: p.grad for k, p in ft.items() if torch.is_tensor(p) and p.requires_grad and p.grad is not None and k not in frozen}
    return dict(loss=loss.item(), loss_ce=ce.item(), loss_temporal=trip.item()), grads
