"""One iteration of the alternating min-max anonymizer training on MI355X: the build's
counterpart of `train_epoch` (anonymization_training/train_anonymizer.py:32-212).

    phase 1 (even iterations, :71-132): fa.train(), ft.eval()
        loss_fa = -fb_loss_weight * NTXent(fb(fa(v0)), fb(fa(v1))) + ft_loss_weight * (CE + temporal_loss_weight * Triplet)
        gradients flow through the frozen ft into fa; optimizer_fa.step()
    phase 2 (odd iterations, :135-198): fa.eval() under no_grad, ft.train()
        loss_ft = CE + temporal_loss_weight * Triplet ; optimizer_ft.step()   (three train-mode ft forwards, Q14)

Video batch layout as the reference loader delivers it: (B, 48, 3, H, W) fp32 in [0,1] = 3 clips x 16
frames stacked on dim 1 (ucf101_dl.py:368-379); labels int64 (B,). The feed reproduces quirk Q2
(`(B,3,48,H,W).reshape(-1,3,H,W)` pseudo-images, train_anonymizer.py:87-92).

The privacy branch fb (ResNet-50 + MLP, resnet50.py) is optional: with `fb_model` and the two
VISPR views (`inputs_vispr`, 2 x (N,3,H,W)) the step is the whole `train_epoch` body; with `fb_model=None` it
optimises the utility term alone and says so in its result (`loss_fb` None).

Gradient range: activations AND activation gradients are stored in f16 (fp32 accumulate). The reference's
train_anonymizer.py back-propagates its fp16-autocast graph without a GradScaler; its action-training scripts use one
(train_anonymized_action.py:92-94). `loss_scale` multiplies the loss gradients entering the networks and is divided out of
every parameter gradient before the optimizer step (powers of two: the same arithmetic); a step whose gradients are not
finite is skipped and reported (`skipped: True`), like `GradScaler.step`. The default is 256, not 1: gfx950's matrix cores
flush f16 SUBNORMAL operands, and the per-pixel activation gradients of the UNet's 112 x 112 / 224 x 224 levels are below
6e-5 at the reference's scale -- the data- and weight-gradient MFMAs dropped them (3-7 % error in the neighbouring
parameter gradients against autograd at the same forward point, 0.5-0.8 % with the scale;
tests/test_hip_train_golden.py::test_phase1_backward_at_the_devices_forward_point_vs_autograd). `loss_scale=1.0` is the
reference's literal behaviour.

Data parallel: one process per GPU, per-rank BatchNorm statistics (what nn.DataParallel does in the
reference). The gradients of the network being updated live in flat buckets (grad_reduce.GradBucketReducer): on
the LAST backward pass of a step every bucket is all-reduced (RCCL) as soon as the backward sequence has finished its stage,
overlapped with the backward of the earlier stages; the optimizer then reads the averaged buckets in place.
"""
from __future__ import annotations

from types import SimpleNamespace

import os

import torch
import torch.distributed as dist

from .grad_reduce import GradBucketReducer
from .losses import CrossEntropyLoss, NTXentLoss, TripletMarginLoss
from . import engine as E
from . import train_engine as TE
from .train_nets import FBTrainer, I3DTrainer, UNetPPTrainer, UNetTrainer

# anonymization_training/params_anonymization.py:28-62
DEFAULT_PARAMS = SimpleNamespace(num_frames=16, learning_rate=1e-5, learning_rate_fa=0.4e-5, learning_rate_fb=1e-5,
                                 learning_rate_ft=1e-5, ft_loss_weight=0.7, fb_loss_weight=1.0, temporal_loss_weight=0.1,
                                 triplet_loss_margin=1, loss="ce", temporal_loss="trip", batch_size=8, batch_size_vispr=12)


def allreduce_mean_grads(params, group=None):
    """Average .grad over the ranks with one flat all-reduce after the whole backward (the round-1 exchange; kept as the reference
    the bucketed reducer is tested against, and for callers that own their gradients)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    grads = [p.grad for p in params if p.grad is not None]
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat /= dist.get_world_size(group)
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()


def ntxent_from_embeddings(z0, z1, temperature=0.1):
    """loss_fb of train_anonymizer.py:82-84 given the two views' fb embeddings (N,128)."""
    return NTXentLoss(z0.device, z0.shape[0], temperature, False)(z0, z1)


def _flag(v):
    """`skipped` always has one type: constants are wrapped too (bool(), ==, repr work the same; json: use bool(flag))."""
    return v if isinstance(v, DeviceFlag) else DeviceFlag.const(bool(v))


class DeviceFlag:
    """Truth value of a 0-d device tensor, read when somebody asks for it (`bool()`, `==`, `repr`): the `skipped` entry of a step result. The non-finite check
    runs on the step's FINAL gradients, so reading it waits for the whole backward pass; a caller that never looks never waits (torch.amp.GradScaler does not
    report a skipped step at all)."""
    __slots__ = ("_t", "_v")

    def __init__(self, t):
        self._t, self._v = t, None

    @classmethod
    def const(cls, v: bool):
        f = cls(None)
        f._v = bool(v)
        return f

    def __bool__(self):
        if self._v is None:
            self._v = bool(self._t.item() != 0)
            self._t = None
        return self._v

    def __eq__(self, other):
        return bool(self) == bool(other)

    def __hash__(self):
        return hash(bool(self))

    def __repr__(self):
        return "DeviceFlag(%s)" % bool(self)


class AnonymizerTrainStep:
    def __init__(self, fa_model, ft_model, params=DEFAULT_PARAMS, fb_model=None, group=None, loss_scale: float = 256.0):
        self.fa, self.ft, self.fb, self.params, self.group = fa_model, ft_model, fb_model, params, group
        self.loss_scale = float(loss_scale)
        E.apply_env_determinism()
        self.batch_clips = os.environ.get("TEDSPAD_TRAIN_BATCH_CLIPS", "1") != "0"   # the three clips of an iteration as one ft batch (0: three passes, A/B)
        self.batch_views = os.environ.get("TEDSPAD_TRAIN_BATCH_VIEWS", "1") != "0"   # the two VISPR views as one fa / fb batch with per-view statistics (0: two passes)
        self.lazy_losses = os.environ.get("TEDSPAD_TRAIN_LAZY_LOSSES", "0") == "1"
        self._pin, self._pin_used, self._posted = None, 0, []         # loss read-back (_post / _collect)
        from .unetpp import UnetPlusPlus
        self.fa_tr, self.ft_tr = (UNetPPTrainer if isinstance(fa_model, UnetPlusPlus) else UNetTrainer)(fa_model), I3DTrainer(ft_model)
        self.fb_tr = FBTrainer(fb_model) if fb_model is not None else None
        self._fa_off_path = self.fa_tr.off_path_params() if hasattr(self.fa_tr, "off_path_params") else []   # unet++: encoder.layer4 (never run)
        def adam(m, lr):          # torch.optim.Adam as train_anonymizer.py:377-380 builds it; the fused implementation where it exists (same update, one launch per
            try:                  # parameter group, and it takes the loss scale's non-finite flag on the device: no host sync before the step)
                return torch.optim.Adam(m.parameters(), lr=lr, fused=True)
            except (RuntimeError, TypeError, ValueError):
                return torch.optim.Adam(m.parameters(), lr=lr)
        self.opt_fa = adam(fa_model, params.learning_rate_fa)
        self.opt_ft = adam(ft_model, params.learning_rate_ft)
        self.opt_fb = adam(fb_model, params.learning_rate_fb) if fb_model is not None else None
        self.ce = CrossEntropyLoss()
        self.trip = TripletMarginLoss(margin=params.triplet_loss_margin)
        self.iteration = 0
        # gradient buckets in the order each backward sequence finishes them (train_nets.*.grad_buckets)
        self.red_fa = GradBucketReducer(self.fa_tr.grad_buckets(), group, all_params=list(fa_model.parameters()))
        self.red_ft = GradBucketReducer(self.ft_tr.grad_buckets(), group, all_params=list(ft_model.parameters()))
        self.red_fb = GradBucketReducer(self.fb_tr.grad_buckets(), group, all_params=list(fb_model.parameters())) if self.fb_tr is not None else None
        # freeze_bn (train_anonymized_action.py:39-40): gamma / beta of the trunk's BatchNorm3d layers are buffers there -> no gradient
        self._frozen_bn_params = [p for n_, p in ft_model.named_parameters()
                                  if n_.startswith("i3d.") and (".bn" in n_ or n_.startswith("i3d.bn") or ".downsample.1." in n_)]

    # ---- shared pieces -------------------------------------------------------------------------------------------
    @staticmethod
    def _feed(inputs_video):
        """(B,48,3,H,W) -> the (B*48,3,H,W) pseudo-image batch fa sees (Q2) and the shape to restore."""
        v = inputs_video.permute(0, 2, 1, 3, 4)                       # :57
        b, c, t, h, w = v.shape
        return v.reshape(-1, c, h, w), (b, c, t, h, w)                # :89 (copy: the permuted tensor is not viewable)

    def _views_batchable(self, views) -> bool:
        """The two VISPR views can run as one batch with per-view BatchNorm statistics: same shape, and every BatchNorm of fa (down to H/16) and fb (down to
        H/32) sees >= 256 values per channel and view (a conv tile of 256 output rows straddles at most one statistics-group boundary)."""
        if not self.batch_views or len(views) != 2 or views[0].shape != views[1].shape:
            return False
        nb, _, h, w = views[0].shape
        return nb * (h // 32) * (w // 32) >= 256 and h % 32 == 0 and w % 32 == 0

    def _utility_losses(self, heads, labels):
        """heads: [(pred, feat)] x3 as leaf tensors -> (loss_ft, loss_ce, loss_trip)."""
        p = self.params
        loss_ce = self.ce(heads[0][0], labels)                        # :107
        loss_trip = self.trip(heads[0][1], heads[1][1], heads[2][1])  # :115
        return loss_ce + p.temporal_loss_weight * loss_trip, loss_ce, loss_trip

    def _post(self, losses: dict):
        """Start reading this step's loss values back NOW: they exist once the forward pass and the loss kernels are queued, long before the backward pass behind
        them has run. One stack + one non-blocking copy into pinned host memory + an event; `_collect` waits for THAT event. (The first version called
        `float(loss)` at the end of the step: a full device sync per phase -- 37 of phase 1's 50 ms with the host idle, then a host-bound phase 2 on an empty
        queue: `scripts/phase_prologue_probe.py`; 81 ms per cfg3 iteration against 72 with the losses left on the device.) `lazy_losses`: nothing is copied,
        the result carries the detached 0-d tensors."""
        live = {k: v.detach() for k, v in losses.items() if v is not None}
        if self.lazy_losses or not live:
            self._posted.append((live, None, 0))
            return
        names = list(live)
        if self._pin is None:
            self._pin = torch.empty(32, dtype=torch.float32, pin_memory=True)
        off = self._pin_used
        self._pin_used += len(names)
        self._pin[off:off + len(names)].copy_(torch.stack([live[k].float().reshape(()) for k in names]), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._posted.append((names, ev, off))

    def _collect(self) -> dict:
        """name -> Python float (or 0-d tensor under `lazy_losses`) of everything `_post` was given in this step."""
        out = {}
        for names, ev, off in self._posted:
            if ev is None:
                out.update(names)
            else:
                ev.synchronize()
                out.update(zip(names, self._pin[off:off + len(names)].tolist()))
        self._posted, self._pin_used = [], 0
        return out

    def _scaled(self, g):
        return g if (g is None or self.loss_scale == 1.0) else g * self.loss_scale

    def _unscale(self, module):
        """Divide the loss scale out of `module`'s gradients. Returns the 0-d device flag "some gradient was not finite" (None without a loss scale).
        Called on the ALL-REDUCED gradients: an overflow on any rank makes the sum non-finite on every rank, so all
        ranks take the same step / skip decision (replicas and Adam step counts stay in lock-step)."""
        if self.loss_scale == 1.0:
            return None
        grads = [p.grad for p in module.parameters() if p.grad is not None]
        found_inf = torch.zeros((), device=grads[0].device)          # 0-d, as torch.amp.GradScaler's (the fused optimizer subtracts it from its 0-d step counters)
        torch._amp_foreach_non_finite_check_and_unscale_(grads, found_inf, torch.full((), 1.0 / self.loss_scale, device=grads[0].device))
        return found_inf

    def _opt_step(self, opt, found_inf, module):
        """optimizer.step(), skipped when `found_inf` is set. With a fused optimizer (the default Adam here) the skip is decided ON THE DEVICE, as torch.amp.GradScaler
        does it: no host sync between the backward pass and the next phase's launches. Returns what goes into the result's `skipped`."""
        fused = getattr(opt, "_step_supports_amp_scaling", False)
        try:
            if found_inf is None:
                opt.step()
                return _flag(False)
            if fused:
                opt.grad_scale, opt.found_inf = None, found_inf
                try:
                    opt.step()
                finally:
                    del opt.grad_scale, opt.found_inf
                return found_inf != 0 if self.lazy_losses else DeviceFlag(found_inf)
            if float(found_inf) != 0.0:
                return _flag(True)
            opt.step()
            return _flag(False)
        finally:
            if fused:       # a fused step leaves the parameters' version counters alone: say that they changed (TE.mark_updated)
                TE.mark_updated(module.parameters())

    @staticmethod
    def _check_deterministic():
        """Deterministic mode (engine.set_deterministic): a workgroup that gave up waiting for its turn opened the gate for the rest of the launch -- that step is
        not reproducible. The give-up counter is sticky until the mode is set again, so it is surfaced here, once per step, as an error (one device read: this
        mode is for tests and debugging)."""
        if E.DETERMINISTIC and E.deterministic_giveups() > 0:
            raise RuntimeError("deterministic mode: %d workgroup(s) gave up their ordered turn in an atomic section during this step; "
                               "its result is not reproducible (call engine.set_deterministic(True) again to re-arm)" % E.deterministic_giveups())

    def _opts(self):
        return [o for o in (self.opt_fa, self.opt_fb, self.opt_ft) if o is not None]

    def _views(self, inputs_vispr):
        if self.fb is None:
            return None
        if inputs_vispr is None or len(inputs_vispr) != 2:
            raise ValueError("a step with fb_model needs inputs_vispr = [view0, view1], each (N,3,H,W) (train_anonymizer.py:56)")
        return inputs_vispr

    # ---- phase 1 --------------------------------------------------------------------------------------------------
    def step_fa(self, inputs_video, labels, inputs_vispr=None):
        """Update fa (phase 1). Returns a dict of python floats (read back as soon as the losses exist, `_post`) + `skipped`, a `DeviceFlag`."""
        self._posted, self._pin_used = [], 0           # (a step that raised between _post and _collect leaves nothing behind)
        p = self.params
        views = self._views(inputs_vispr)
        self.fa.train(); self.ft.eval()
        if self.fb is not None:
            self.fb.eval()
        for opt in self._opts():
            opt.zero_grad(set_to_none=True)
        TE.ARENA.reset(inputs_video.device)
        self.red_fa.prepare(exclude=self._fa_off_path)                # fa's gradients: zeroed views into the buckets
        fb_ctx, loss_fb = [], None
        if views is not None and self._views_batchable(views):        # :80-84 as ONE batch: fa's BatchNorms keep the statistics of each view (groups = 2), fb is frozen
            nb = views[0].shape[0]
            y, tape_u = self.fa_tr.forward(torch.cat(views, dim=0), groups=2)
            emb, tape_b = self.fb_tr.forward(y, "eval")
            z = emb.detach().requires_grad_()
            fb_ctx.append((tape_u, tape_b, z))
            loss_fb = NTXentLoss(inputs_video.device, nb, 0.1, False)(z[:nb], z[nb:])
        elif views is not None:                                       # :80-84: fa (train mode) on each view, frozen fb
            for v in views:
                y, tape_u = self.fa_tr.forward(v)
                emb, tape_b = self.fb_tr.forward(y, "eval")
                fb_ctx.append((tape_u, tape_b, emb.detach().requires_grad_()))
            loss_fb = NTXentLoss(inputs_video.device, fb_ctx[0][2].shape[0], 0.1, False)(fb_ctx[0][2], fb_ctx[1][2])
        frames, shape = self._feed(inputs_video)
        anon_flat, tape_fa = self.fa_tr.forward(frames)
        anon = anon_flat.reshape(shape)                               # :92
        clips = torch.split(anon, [p.num_frames] * 3, dim=2)          # :94 (non-contiguous views, Q15)
        tapes, leaves = [], []
        if self.batch_clips:          # ft is frozen here (BatchNorm folded): the three clips are independent samples of ONE batch
            nb = clips[0].shape[0]
            pred, feat, tape3 = self.ft_tr.forward(torch.cat(clips, dim=0), "eval")
            P3, F3 = pred.detach().requires_grad_(), feat.detach().requires_grad_()
            leaves = [(P3[k * nb:(k + 1) * nb], F3[k * nb:(k + 1) * nb]) for k in range(3)]
        else:
            for c in clips:
                pred, feat, tape = self.ft_tr.forward(c, "eval")
                tapes.append(tape)
                leaves.append((pred.detach().requires_grad_(), feat.detach().requires_grad_()))
        loss_ft, loss_ce, loss_trip = self._utility_losses(leaves, labels)
        loss_fa = p.ft_loss_weight * loss_ft                          # :119
        if loss_fb is not None:
            loss_fa = -p.fb_loss_weight * loss_fb + loss_fa
        self._post(dict(loss_fa=loss_fa, loss_ft=loss_ft, loss_ce=loss_ce, loss_temporal=loss_trip, loss_fb=loss_fb))
        loss_fa.backward()
        for tape_u, tape_b, z in fb_ctx:
            self.fa_tr.backward(tape_u, self.fb_tr.backward(tape_b, self._scaled(z.grad)))
        if self.batch_clips:
            d3 = self.ft_tr.backward(tape3, self._scaled(P3.grad), self._scaled(F3.grad))                    # (3B,3,16,H,W)
            danon = torch.cat(torch.split(d3, d3.shape[0] // 3, dim=0), dim=2)                              # (B,3,48,H,W)
        else:
            danon = torch.zeros(shape, dtype=torch.float32, device=anon.device)
            for k, (tape, (pl, fl)) in enumerate(zip(tapes, leaves)):
                self.ft_tr.backward(tape, self._scaled(pl.grad), self._scaled(fl.grad), dx_out=danon[:, :, k * p.num_frames:(k + 1) * p.num_frames])
        self.fa_tr.backward(tape_fa, danon.reshape(anon_flat.shape), on_bucket_done=self.red_fa.bucket_ready)   # fa's last backward pass of the step
        self.fa_tr.flush_grads()
        self.red_fa.finish()
        skipped = self._opt_step(self.opt_fa, self._unscale(self.fa), self.fa)     # :123
        self.iteration += 1
        self._check_deterministic()
        v = self._collect()
        return dict(phase=1, loss_fa=v["loss_fa"], loss_ft=v["loss_ft"], loss_ce=v["loss_ce"], loss_temporal=v["loss_temporal"], loss_fb=v.get("loss_fb"), skipped=skipped)

    def _three_clips(self, clips, labels, mode, drop_masks):
        """Forward + loss + backward of ft ('train' / 'frozen') on the three clips of an iteration (:169-175 / action :64-84). The reference
        calls ft_model once per clip; here the three calls are ONE launch sequence over a batch of three GROUPS whose train-mode
        BatchNorms keep separate batch statistics and update their running statistics group after group (I3DTrainer.forward, groups=3) --
        whenever every BatchNorm of the trunk sees >= 256 values per channel and group; otherwise three passes as before."""
        nb = clips[0].shape[0]
        shape3 = (3 * nb,) + tuple(clips[0].shape[1:])
        if self.batch_clips and (mode == "frozen" or self.ft_tr.min_group_rows(shape3, 3) >= 256):
            dm = None if drop_masks is None else torch.cat([m for m in drop_masks], dim=0)
            x3 = torch.cat(clips, dim=0)

            def body(x, lab):
                pred, feat, tape = self.ft_tr.forward(x, mode, drop_mask=dm, groups=3)
                P3, F3 = pred.detach().requires_grad_(), feat.detach().requires_grad_()
                losses = self._utility_losses([(P3[k * nb:(k + 1) * nb], F3[k * nb:(k + 1) * nb]) for k in range(3)], lab)
                self._post(dict(loss_ft=losses[0], loss_ce=losses[1], loss_temporal=losses[2]))
                losses[0].backward()
                self.ft_tr.backward(tape, self._scaled(P3.grad), self._scaled(F3.grad), on_bucket_done=self.red_ft.bucket_ready)
                return losses

            return body(x3, labels)
        tapes, leaves = [], []
        for k, c in enumerate(clips):
            pred, feat, tape = self.ft_tr.forward(c, mode, drop_mask=None if drop_masks is None else drop_masks[k])
            tapes.append(tape)
            leaves.append((pred.detach().requires_grad_(), feat.detach().requires_grad_()))
        losses = self._utility_losses(leaves, labels)
        self._post(dict(loss_ft=losses[0], loss_ce=losses[1], loss_temporal=losses[2]))
        losses[0].backward()
        for j, (tape, (pl, fl)) in enumerate(zip(tapes, leaves)):     # the third clip's pass finishes every bucket -> all-reduce under it
            self.ft_tr.backward(tape, self._scaled(pl.grad), self._scaled(fl.grad),
                                on_bucket_done=self.red_ft.bucket_ready if j == len(tapes) - 1 else None)
        return losses

    # ---- phase 2 --------------------------------------------------------------------------------------------------
    def step_ft(self, inputs_video, labels, drop_masks=None, inputs_vispr=None):
        """Update ft and fb (phase 2)."""
        self._posted, self._pin_used = [], 0           # (a step that raised between _post and _collect leaves nothing behind)
        p = self.params
        views = self._views(inputs_vispr)
        self.fa.eval(); self.ft.train()
        if self.fb is not None:
            self.fb.train()
        for opt in self._opts():
            opt.zero_grad(set_to_none=True)
        TE.ARENA.reset(inputs_video.device)
        self.red_ft.prepare()
        if self.red_fb is not None:
            self.red_fb.prepare()
        frames, shape = self._feed(inputs_video)
        together = views is not None and self._views_batchable(views)
        with torch.no_grad():
            if together:                                              # fa is in eval mode here: per-sample, the two views are one batch
                anon_views = [self.fa(torch.cat(views, dim=0))]
            else:
                anon_views = [self.fa(v) for v in views] if views is not None else []      # :147
            anon = self.fa(frames).reshape(shape)                     # :144-148
        loss_fb = None
        if views is not None:                                         # :153-157,190,192
            ctx = []
            if together:                                              # fb in train mode on both views at once: separate batch statistics per view (groups = 2)
                nb = views[0].shape[0]
                emb, tape_b = self.fb_tr.forward(anon_views[0], "train", groups=2)
                z = emb.detach().requires_grad_()
                ctx.append((tape_b, z))
                loss_fb = NTXentLoss(inputs_video.device, nb, 0.1, False)(z[:nb], z[nb:])
            else:
                for x in anon_views:
                    emb, tape_b = self.fb_tr.forward(x, "train")
                    ctx.append((tape_b, emb.detach().requires_grad_()))
                loss_fb = NTXentLoss(inputs_video.device, ctx[0][1].shape[0], 0.1, False)(ctx[0][1], ctx[1][1])
            self._post(dict(loss_fb=loss_fb))
            loss_fb.backward()
            for j, (tape_b, z) in enumerate(ctx):
                self.fb_tr.backward(tape_b, self._scaled(z.grad), on_bucket_done=self.red_fb.bucket_ready if j == len(ctx) - 1 else None)
            self.fb_tr.flush_grads()
            self.red_fb.finish()
            self._opt_step(self.opt_fb, self._unscale(self.fb), self.fb)
        clips = torch.split(anon, [p.num_frames] * 3, dim=2)
        loss_ft, loss_ce, loss_trip = self._three_clips(clips, labels, "train", drop_masks)             # :169-175,191
        self.ft_tr.flush_grads()
        self.red_ft.finish()
        skipped = self._opt_step(self.opt_ft, self._unscale(self.ft), self.ft)     # :193
        self.iteration += 1
        self._check_deterministic()
        v = self._collect()
        return dict(phase=2, loss_ft=v["loss_ft"], loss_ce=v["loss_ce"], loss_temporal=v["loss_temporal"], loss_fb=v.get("loss_fb"), skipped=skipped)

    def step_action(self, inputs_video, labels, drop_masks=None):
        """One iteration of action_training/train_anonymized_action.py:43-94 (`--temporal_loss trip`, cross-entropy): the
        anonymizer is frozen and run without gradients (:52-57), ft is trained on the anonymised clips with its trunk
        BatchNorm3d layers frozen (`freeze_bn`, :39-40: running statistics, gamma / beta are buffers and get no gradient;
        dropout and the mlp head follow the train flag), loss = CE(pred of clip 1) + w * triplet(feat1, feat2, feat3)
        (:64-84), then the optimizer step on ft (:86-88; the GradScaler is the static `loss_scale` here)."""
        self._posted, self._pin_used = [], 0           # (a step that raised between _post and _collect leaves nothing behind)
        p = self.params
        self.fa.eval(); self.ft.train()
        self.opt_ft.zero_grad(set_to_none=True)                       # :46
        TE.ARENA.reset(inputs_video.device)
        self.red_ft.prepare(exclude=self._frozen_bn_params)
        frames, shape = self._feed(inputs_video)                      # :47,54-55 (Q2)
        with torch.no_grad():
            anon = self.fa(frames).reshape(shape)                     # :56-57
        clips = torch.split(anon, [p.num_frames] * 3, dim=2)          # :62
        loss, loss_ce, loss_trip = self._three_clips(clips, labels, "frozen", drop_masks)
        self.ft_tr.flush_grads()
        self.red_ft.finish()
        skipped = self._opt_step(self.opt_ft, self._unscale(self.ft), self.ft)     # :87
        self.iteration += 1
        self._check_deterministic()
        v = self._collect()
        return dict(phase="action", loss=v["loss_ft"], loss_ce=v["loss_ce"], loss_temporal=v["loss_temporal"], skipped=skipped)

    def step(self, inputs_video, labels, inputs_vispr=None):
        """Alternates like train_epoch's `step` flag (:71,135): even iterations update fa, odd ones ft (and fb)."""
        if self.iteration % 2 == 0:
            return self.step_fa(inputs_video, labels, inputs_vispr)
        return self.step_ft(inputs_video, labels, inputs_vispr=inputs_vispr)
