"""Forward-with-tape and backward of the three networks of the anonymizer training step on
MI355X: the reference relies on torch autograd through its nn.Modules
(anonymization_training/train_anonymizer.py:71-132 phase 1, :135-198 phase 2); here each network
has an explicit launch sequence for both directions (all arithmetic in libtedspad_hip.so).

  I3DTrainer  -- wrapper_i3d (I3Res50 + fc + mlp):
       mode 'eval'  (phase 1, ft frozen: BN folded, gradient flows to the INPUT only; the unused
                     weight gradients of the reference are skipped)
       mode 'train' (phase 2: batch-statistics BN, running stats updated once per forward (Q14),
                     weight gradients, dropout before fc)
  UNetTrainer -- UNet anonymizer in train mode (phase 1).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib, engine as E, head, train_engine as TE
from ._lib import check
from .engine import Act, _stream_ptr


def _acc_grad(p, g):
    if p.grad is None:
        p.grad = g.clone()
    else:
        p.grad.add_(g)          # in place: p.grad may be a view into a gradient bucket (grad_reduce.GradBucketReducer)


# ------------------------------------------------------------------------------------------------------------------
# fp32 head: Linear / BatchNorm1d(train) / L2-normalise with explicit backward (B x C matrices, B tiny)
# ------------------------------------------------------------------------------------------------------------------

def _linear_bwd(x, w, dy, want_dx=True):
    """y = x w^T (+b): returns (dx, dw, db); the three products reuse the forward GEMV kernel."""
    dx = head.linear(dy, w.detach().t().contiguous()) if want_dx else None            # (B,N) x (N,K)
    dw = head.linear(dy.t().contiguous(), x.t().contiguous())                          # (N,B) x (B,K)
    db = head.linear(torch.ones((1, dy.shape[0]), device=dy.device), dy.t().contiguous())[0]
    return dx, dw, db


def _bn1d_train(x, bn, relu):
    B, Cn = x.shape
    y = torch.empty_like(x)
    mean, invstd = torch.empty(Cn, device=x.device), torch.empty(Cn, device=x.device)
    check(_lib.lib().tedspad_bn1d_train_fwd(x.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), C.c_float(bn.eps), C.c_float(bn.momentum),
                                            bn.running_mean.data_ptr(), bn.running_var.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                            invstd.data_ptr(), B, Cn, int(relu), _stream_ptr()), "tedspad_bn1d_train_fwd")
    TE.bump_counter(bn.num_batches_tracked)
    return y, (x, y, mean, invstd, relu)


def _bn1d_train_bwd(ctx, bn, dy):
    x, y, mean, invstd, relu = ctx
    B, Cn = x.shape
    dy = dy.contiguous()
    dx, dg, db = torch.empty_like(x), torch.empty(Cn, device=x.device), torch.empty(Cn, device=x.device)
    check(_lib.lib().tedspad_bn1d_train_bwd(dy.data_ptr(), x.data_ptr(), y.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                            bn.weight.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), B, Cn, int(relu),
                                            _stream_ptr()), "tedspad_bn1d_train_bwd")
    return dx, dg, db


def _bn1d_train_groups(x, bn, relu, groups):
    """BatchNorm1d in train mode on `groups` consecutive row blocks of x, each with its own batch statistics (the module called once per
    block, in order: the running statistics take one momentum update per block)."""
    if groups == 1:
        y, ctx = _bn1d_train(x, bn, relu)
        return y, [ctx]
    nb = x.shape[0] // groups
    parts = [_bn1d_train(x[g * nb:(g + 1) * nb], bn, relu) for g in range(groups)]
    return torch.cat([p[0] for p in parts]), [p[1] for p in parts]


def _bn1d_train_bwd_groups(ctxs, bn, dy):
    if len(ctxs) == 1:
        return _bn1d_train_bwd(ctxs[0], bn, dy)
    nb = dy.shape[0] // len(ctxs)
    parts = [_bn1d_train_bwd(c, bn, dy[g * nb:(g + 1) * nb]) for g, c in enumerate(ctxs)]
    return torch.cat([p[0] for p in parts]), sum(p[1] for p in parts), sum(p[2] for p in parts)


def _l2norm_bwd(x, dy, eps=1e-12):
    dx = torch.empty_like(x)
    dy = dy.contiguous().float()
    check(_lib.lib().tedspad_l2_normalize_rows_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.shape[0], x.shape[1], C.c_float(eps),
                                                   _stream_ptr()), "tedspad_l2_normalize_rows_bwd")
    return dx


def _mul(a, b, scale):
    out = torch.empty_like(a)
    check(_lib.lib().tedspad_mul_f32(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), C.c_float(scale), _stream_ptr()), "tedspad_mul_f32")
    return out


# ------------------------------------------------------------------------------------------------------------------
# bottleneck trunk shared by wrapper_i3d (I3Res50) and the privacy branch fb (ResNet-50 == the same blocks with
# 2-D kernels: an image is a clip with T = 1)
# ------------------------------------------------------------------------------------------------------------------

class BottleneckTrunk:
    """stem conv + BN + ReLU -> max-pool -> bottleneck blocks (optionally a max-pool in front of a block) -> global
    average pool, forward with tape and both backward flavours:
       'train' : batch-statistics BN, parameter gradients (+ nothing w.r.t. the input)
       'eval'  : BN folded, gradient w.r.t. the INPUT only (the network is frozen in that phase)
       'frozen': BN folded (FrozenBN of large_i3d.py:8-38: running statistics, gamma / beta are buffers), gradients of the CONV
                 WEIGHTS only -- `freeze_bn(ft_model)` of action_training/train_anonymized_action.py:39-40."""

    def __init__(self, stem, stem_bn, pool1, blocks):
        """pool1 = (kernel, stride, pads); blocks = dicts with c1,c2,c3,cd (ConvLayer | None), bn1,bn2,bn3,bnd,
        pre_pool ((kernel, stride) | None)."""
        self.stem, self.stem_bn, self.pool1, self.blocks = stem, stem_bn, pool1, blocks
        self._folds = {}
        self.refresh = TE.WeightRefresh(self.conv_layers, self._folds)

    def conv_layers(self):
        out = [self.stem]
        for d in self.blocks:
            out += [d[k] for k in ("c1", "c2", "c3", "cd") if d[k] is not None]
        return out

    def flush_grads(self):
        """Convert the packed weight-gradient accumulators of this step into the parameters' .grad."""
        TE.flush_conv_grads(self.conv_layers())

    def fold(self, bn):
        """Eval-mode BN as fp32 (scale, shift); cached, re-folded in place when the BN tensors changed."""
        return TE.cached_fold(self._folds, bn)

    def forward(self, a: Act, train: bool, frozen: bool = False, groups: int = 1):
        """a: the pixel-pair input Act. Returns (f (B,C) fp32 pooled feature, tape).
        groups (train mode): the batch is `groups` blocks of samples whose BatchNorms keep separate batch statistics (TE.conv_bn_act_train)."""
        assert not (train and frozen)
        self.refresh.run()              # after an optimizer step: every existing weight image / BatchNorm fold rewritten in place, two launches
        tape = dict(train=train, frozen=frozen, units=[], clip=a)

        def unit(conv, bn, xin, relu=True, residual=None):
            if train:
                return TE.conv_bn_act_train(conv, bn, xin, relu=relu, residual=residual, groups=groups)
            s, b = self.fold(bn)
            return conv.forward(xin, scale=s, shift=b, relu=relu, residual=residual), (conv, s)

        y, tape["stem"] = unit(self.stem, self.stem_bn, a)
        tape["stem_y"] = y
        k, s, pads = self.pool1
        a, tape["idx1"] = E.maxpool(y, k, s, pads=pads, return_idx=True)
        after_pool = True
        for d in self.blocks:
            a, rec = self.block_forward(d, a, unit, after_pool)
            after_pool = False
            tape["units"].append(rec)
        f = E.global_avgpool(a)
        tape["f"] = f
        if train:
            TE.flush_counters()
        return f, tape

    @staticmethod
    def block_forward(d, a: Act, unit, after_pool: bool):
        """One bottleneck (large_i3d.py:61-84; with `pre_pool`, maxpool2 in front of it): conv1-bn1-relu, conv2-bn2-relu, conv3-bn3 (+ downsample
        branch | identity) - relu. `unit(conv, bn, x, relu=, residual=)` is the conv+BN(+residual)+ReLU launch pair of the current mode.
        Returns (block output, tape record)."""
        rec = dict(d=d)
        if d["pre_pool"] is not None:
            rec["pool_in"] = a
            a, rec["pool_idx"] = E.maxpool(a, d["pre_pool"][0], d["pre_pool"][1], return_idx=True)
            after_pool = True
        rec["a_in"], rec["after_pool"] = a, after_pool           # after_pool: the block input is a max-pool output, not a ReLU output
        h1, rec["u1"] = unit(d["c1"], d["bn1"], a)
        h2, rec["u2"] = unit(d["c2"], d["bn2"], h1)
        if d["cd"] is not None:
            r, rec["ud"] = unit(d["cd"], d["bnd"], a, relu=False)
            rec["r"] = r
        else:
            r = a
        a, rec["u3"] = unit(d["c3"], d["bn3"], h2, relu=True, residual=r)
        rec["h1"], rec["h2"], rec["out"] = h1, h2, a
        return a, rec

    @staticmethod
    def block_backward_train(rec, da: Act) -> Act:
        """Train-mode backward of one bottleneck: da = gradient w.r.t. the block output; accumulates the block's parameter gradients and
        returns the gradient w.r.t. the block input (in front of `pre_pool` if the block has one)."""
        dh2, dres = TE.conv_bn_act_train_bwd(rec["u3"], da)
        dh1, _ = TE.conv_bn_act_train_bwd(rec["u2"], dh2)
        t = TE.conv_bn_act_train_bwd(rec["ud"], dres)[0] if "ud" in rec else dres
        da, _ = TE.conv_bn_act_train_bwd(rec["u1"], dh1, dx_residual=t)
        if "pool_idx" in rec:
            pk, ps = rec["d"]["pre_pool"]
            da = TE.maxpool_bwd(rec["pool_in"], rec["pool_idx"], da, pk, ps)
        return da

    def stage_params(self):
        """Parameter groups in the order their gradients become final in `backward`: [layer4, layer3, layer2, layer1, stem]."""
        groups = {}
        for d in self.blocks:
            g = groups.setdefault(d["li"], [])
            for c, bn in ((d["c1"], d["bn1"]), (d["c2"], d["bn2"]), (d["c3"], d["bn3"]), (d["cd"], d["bnd"])):
                if c is not None:
                    g += [c.weight] + ([c.bias] if c.bias is not None else []) + [bn.weight, bn.bias]
        out = [groups[li] for li in sorted(groups, reverse=True)]
        out.append([self.stem.weight] + ([self.stem.bias] if self.stem.bias is not None else []) + [self.stem_bn.weight, self.stem_bn.bias])
        return out

    def _flush_stage(self, li):
        TE.flush_conv_grads([d[k] for d in self.blocks if d["li"] == li for k in ("c1", "c2", "c3", "cd") if d[k] is not None])

    def backward(self, tape, df: torch.Tensor, on_stage_done=None):
        """df: (B,C) fp32 gradient w.r.t. the pooled feature. 'train' tape: accumulates the parameter gradients,
        returns None. 'eval' tape: returns d(input) as the stem's pixel-pair Act.
        on_stage_done(k): called (train / frozen tapes) right after the k-th parameter group of `stage_params()` received its last
        contribution of this pass and was flushed into `.grad` -- the hook the bucketed gradient all-reduce hangs on."""
        last = tape["units"][-1]["out"]
        k1, s1, p1 = self.pool1
        stages = sorted({d["li"] for d in self.blocks}, reverse=True)
        if tape["train"]:
            da = TE.global_avgpool_bwd(df, last)
            for rec in reversed(tape["units"]):
                da = self.block_backward_train(rec, da)
                if on_stage_done is not None and rec["d"]["bi"] == 0:       # first block of a stage = the last one backward reaches
                    self._flush_stage(rec["d"]["li"])
                    on_stage_done(stages.index(rec["d"]["li"]))
            da = TE.maxpool_bwd(tape["stem_y"], tape["idx1"], da, k1, s1, pads=p1)
            TE.conv_bn_act_train_bwd(tape["stem"], da, need_dx=False)
            if on_stage_done is not None:
                self.stem.flush_grad()
                on_stage_done(len(stages))
            return None
        # eval / frozen mode. delta = gradient w.r.t. a block output's PRE-activation (already ReLU-masked).
        # frozen: every conv also gets its weight gradient: d(conv output) = delta * scale per output channel, so the packed
        # accumulator takes wgrad(x, delta) and the rows are multiplied by the folded BN scale when it is flushed.
        frozen = tape.get("frozen", False)

        def wg(unit, x, d):
            if frozen:
                conv, sc = unit
                conv.wgrad(x, d)
                conv._grad_row_scale = sc

        delta = TE.global_avgpool_bwd(df, last, mask=last)
        for rec in reversed(tape["units"]):
            (c3, s3), (c2, s2), (c1, sc1) = rec["u3"], rec["u2"], rec["u1"]
            wg(rec["u3"], rec["h2"], delta)
            du2 = c3.dgrad(delta, rec["h2"].dims[1:], scale=s3, mask=rec["h2"])
            wg(rec["u2"], rec["h1"], du2)
            du1 = c2.dgrad(du2, rec["h1"].dims[1:], scale=s2, mask=rec["h1"])
            wg(rec["u1"], rec["a_in"], du1)
            if "ud" in rec:
                wg(rec["ud"], rec["a_in"], delta)
            t = rec["ud"][0].dgrad(delta, rec["a_in"].dims[1:], scale=rec["ud"][1]) if "ud" in rec else delta
            delta = c1.dgrad(du1, rec["a_in"].dims[1:], scale=sc1, residual=t, mask=None if rec["after_pool"] else rec["a_in"])
            if "pool_idx" in rec:
                pk, ps = rec["d"]["pre_pool"]
                delta = TE.maxpool_bwd(rec["pool_in"], rec["pool_idx"], delta, pk, ps, relu_mask=True)
        delta = TE.maxpool_bwd(tape["stem_y"], tape["idx1"], delta, k1, s1, pads=p1, relu_mask=True)
        conv, s = tape["stem"]
        if frozen:                                                           # the anonymizer in front is not trained: no d(input)
            wg(tape["stem"], tape["clip"], delta)
            if on_stage_done is not None:
                self.flush_grads()
                for k in range(len(stages) + 1):
                    on_stage_done(k)
            return None
        return conv.dgrad(delta, tape["clip"].dims[1:], scale=s)             # (B,T,H,W/2,8) == (B,T,H,W,4)


def _bottleneck_blocks(layers, dt, temporal):
    """ConvLayers of a list of `nn.Sequential` bottleneck stages (parameter names conv1..3 / bn1..3 / downsample)."""
    blocks = []
    for li, layer in enumerate(layers, 1):
        for bi, blk in enumerate(layer):
            s = blk.stride
            tc = blk.temp_conv if temporal else 0
            blocks.append(dict(
                li=li, bi=bi, bn1=blk.bn1, bn2=blk.bn2, bn3=blk.bn3, bnd=blk.downsample[1] if blk.downsample is not None else None,
                c1=TE.ConvLayer(blk.conv1.weight, None, (1, 1, 1), (tc, 0, 0), dtype=dt),
                c2=TE.ConvLayer(blk.conv2.weight, None, (1, s, s), (0, 1, 1), dtype=dt),
                c3=TE.ConvLayer(blk.conv3.weight, None, (1, 1, 1), (0, 0, 0), dtype=dt),
                cd=TE.ConvLayer(blk.downsample[0].weight, None, (1, s, s), (0, 0, 0), dtype=dt) if blk.downsample is not None else None,
                pre_pool=None))
    return blocks


# ------------------------------------------------------------------------------------------------------------------
# wrapper_i3d
# ------------------------------------------------------------------------------------------------------------------

class I3DTrainer:
    def __init__(self, wrapper):
        self.m = wrapper
        i3d = wrapper.i3d
        dt = i3d.compute_dtype
        stem = TE.ConvLayer(i3d.conv1.weight, None, (2, 2, 2), (2, 3, 3), pair_w=3, dtype=dt)
        blocks = _bottleneck_blocks([getattr(i3d, "layer%d" % li) for li in range(1, 5)], dt, temporal=True)
        for d in blocks:
            if (d["li"], d["bi"]) == (2, 0):
                d["pre_pool"] = ((2, 1, 1), (2, 1, 1))                            # large_i3d.py:139
        self.trunk = BottleneckTrunk(stem, i3d.bn1, ((2, 3, 3), (2, 2, 2), (0, 0, 0)), blocks)     # large_i3d.py:138
        self.stem, self.blocks = stem, blocks

    def conv_layers(self):
        return self.trunk.conv_layers()

    def flush_grads(self):
        self.trunk.flush_grads()

    def _fold(self, bn):
        return self.trunk.fold(bn)

    def min_group_rows(self, x_shape, groups: int) -> int:
        """Rows (samples x output positions) per statistics group at the SMALLEST BatchNorm of the trunk (layer4) for an input of
        `x_shape` split into `groups` blocks: grouped statistics need >= 256 of them (one tile straddles at most one group boundary)."""
        B, _, t, h, w = x_shape
        co = E.conv_out
        st = self.stem
        (kt, kh, kw), (s0, s1, s2), (p0, p1, p2) = tuple(st._w5().shape[2:]), st.stride, st.pads
        t, h, w = co(t, kt, s0, p0, p0), co(h, kh, s1, p1, p1), co(w, kw, s2, p2, p2)
        (k, s, pd) = self.trunk.pool1
        t, h, w = co(t, k[0], s[0], pd[0], pd[0]), co(h, k[1], s[1], pd[1], pd[1]), co(w, k[2], s[2], pd[2], pd[2])
        for d in self.blocks:
            if d["pre_pool"] is not None:
                (k, s) = d["pre_pool"]
                t, h, w = co(t, k[0], s[0], 0, 0), co(h, k[1], s[1], 0, 0), co(w, k[2], s[2], 0, 0)
            s2_ = d["c2"].stride
            h, w = co(h, 3, s2_[1], 1, 1), co(w, 3, s2_[2], 1, 1)
        return (B // groups) * t * h * w

    # ---- forward ---------------------------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, mode: str, drop_mask: Optional[torch.Tensor] = None, groups: int = 1):
        """x: (B,3,T,H,W) fp32 (any strides: a `torch.split` view is fine).
        Returns (pred (B,nc), feat (B,128), tape).
        groups > 1: x holds `groups` batches of B / groups clips that the reference passes through the module in SEPARATE calls (the three
        clips of an iteration, train_anonymizer.py:169-175): every train-mode BatchNorm (3d in the trunk, 1d in the mlp head) normalises each
        block with its own batch statistics and updates its running statistics once per block, in order -- the same arithmetic as the
        separate calls, a third of the launches, three times the work per launch."""
        assert mode in ("eval", "train", "frozen")
        i3d, mlp = self.m.i3d, self.m.mlp
        E.require_cuda(x, "I3DTrainer")
        if x.shape[0] < 2 * groups or x.shape[0] % groups:
            raise ValueError("wrapper_i3d.forward needs B >= 2 (BatchNorm1d; SURVEY.md Q3)")
        # 'frozen' (train_anonymized_action.py:39-40): freeze_bn swaps only the BatchNorm3d modules of the trunk; dropout and the
        # mlp head's BatchNorm1d keep following the module's train flag, i.e. behave as in 'train'
        train = mode in ("train", "frozen")
        f, tape = self.trunk.forward(E.clip_to_act(x, cpad=4, dtype=i3d.compute_dtype), mode == "train", frozen=mode == "frozen",
                                     groups=groups if mode == "train" else 1)                  # feat = x.squeeze() BEFORE dropout
        tape["mode"], tape["x_shape"], tape["groups"] = mode, tuple(x.shape), groups
        # ---- head: fc on dropout(f) ; mlp on f ---------------------------------------------------------------------
        fd = f
        if train and i3d.drop_p > 0:
            if drop_mask is None:
                drop_mask = (torch.rand_like(f) >= i3d.drop_p).float()
            fd = _mul(f, drop_mask, 1.0 / (1.0 - i3d.drop_p))
        tape["drop_mask"], tape["fd"] = drop_mask, fd
        pred = head.linear(fd, i3d.fc.weight, i3d.fc.bias)
        if train:
            z1 = head.linear(f, mlp.fc1.weight, mlp.fc1.bias)
            h, tape["bn1"] = _bn1d_train_groups(z1, mlp.bn1, True, groups)
            z2 = head.linear(h, mlp.fc2.weight, None)
            g, tape["bn2"] = _bn1d_train_groups(z2, mlp.bn2, False, groups)
            tape["h"], tape["g"] = h, g
            feat = head.l2_normalize(g)
        else:
            h = head.linear(f, mlp.fc1.weight, mlp.fc1.bias, bn=mlp.bn1, relu=True)
            g = head.linear(h, mlp.fc2.weight, None, bn=mlp.bn2, relu=False)
            tape["h"], tape["g"] = h, g
            feat = head.l2_normalize(g)
        return pred, feat, tape

    # ---- backward --------------------------------------------------------------------------------------------------
    def grad_buckets(self):
        """Parameters grouped in the order `backward` finishes them: [head (fc + mlp), layer4, layer3, layer2, layer1, stem]."""
        i3d, mlp = self.m.i3d, self.m.mlp
        head = [i3d.fc.weight, i3d.fc.bias] + list(mlp.parameters())
        return [head] + self.trunk.stage_params()

    def backward(self, tape, dpred: Optional[torch.Tensor], dfeat: Optional[torch.Tensor], dx_out: Optional[torch.Tensor] = None,
                 on_bucket_done=None):
        """Accumulates parameter gradients (mode 'train') and/or writes d(loss)/d(clip) into `dx_out`
        ((B,3,T,H,W) fp32 view, any strides; mode 'eval'). on_bucket_done(k): the k-th group of `grad_buckets()` is final
        (pass it on the LAST backward pass of a step only: earlier passes still add to every group)."""
        i3d, mlp = self.m.i3d, self.m.mlp
        train = tape["mode"] in ("train", "frozen")
        f, h, g = tape["f"], tape["h"], tape["g"]
        df = torch.zeros_like(f)
        if dpred is not None:
            dpred = dpred.contiguous().float()
            dfd, dw, db = _linear_bwd(tape["fd"], i3d.fc.weight, dpred)
            if train:
                _acc_grad(i3d.fc.weight, dw)
                _acc_grad(i3d.fc.bias, db)
                if tape["drop_mask"] is not None and i3d.drop_p > 0:
                    dfd = _mul(dfd, tape["drop_mask"], 1.0 / (1.0 - i3d.drop_p))
            df = df + dfd
        if dfeat is not None:
            dg = _l2norm_bwd(g, dfeat)
            if train:
                dz2, dgam, dbet = _bn1d_train_bwd_groups(tape["bn2"], mlp.bn2, dg)
                _acc_grad(mlp.bn2.weight, dgam); _acc_grad(mlp.bn2.bias, dbet)
                dh, dw2, _ = _linear_bwd(h, mlp.fc2.weight, dz2)
                _acc_grad(mlp.fc2.weight, dw2)
                dz1, dgam, dbet = _bn1d_train_bwd_groups(tape["bn1"], mlp.bn1, dh)
                _acc_grad(mlp.bn1.weight, dgam); _acc_grad(mlp.bn1.bias, dbet)
                dfm, dw1, db1 = _linear_bwd(f, mlp.fc1.weight, dz1)
                _acc_grad(mlp.fc1.weight, dw1); _acc_grad(mlp.fc1.bias, db1)
            else:
                s2, _ = self._fold(mlp.bn2)
                dz2 = dg * s2
                dh = head.linear(dz2, mlp.fc2.weight.detach().t().contiguous())
                s1, _ = self._fold(mlp.bn1)
                dz1 = _mul(dh, (h > 0).float(), 1.0) * s1
                dfm = head.linear(dz1, mlp.fc1.weight.detach().t().contiguous())
            df = df + dfm
        if train and on_bucket_done is not None:
            on_bucket_done(0)                                   # fc + mlp gradients are complete
        dclip = self.trunk.backward(tape, df, on_stage_done=None if on_bucket_done is None else (lambda k: on_bucket_done(k + 1)))
        if train:
            return None
        B, _, T, Hh, Ww = tape["x_shape"]
        dview = Act(dclip.buf.view(B, T, Hh, Ww, 4), 4)
        if dx_out is None:
            dx_out = torch.empty(tape["x_shape"], dtype=torch.float32, device=dclip.buf.device)
        TE.act_to_nchw_into(dview, 3, dx_out)
        return dx_out


# ------------------------------------------------------------------------------------------------------------------
# fb: the privacy branch, ResNet-50 + MLP (model_loaders.py:124-153)
# ------------------------------------------------------------------------------------------------------------------

class FBTrainer:
    """nn.Sequential(ResNet50(fc = Identity), MLP(2048 -> 2048 -> 128, L2-normalised)) with tape:
       mode 'eval'  (phase 1, train_anonymizer.py:75-84: fb frozen, the NT-Xent gradient flows through it into fa)
       mode 'train' (phase 2, :138-157,190-192: fb is updated with the NT-Xent loss of the anonymised views)."""

    def __init__(self, fb_model):
        self.m = fb_model
        r = fb_model[0]
        dt = r.compute_dtype
        stem = TE.ConvLayer(r.conv1.weight, None, (1, 2, 2), (0, 3, 3), pair_w=3, dtype=dt)
        blocks = _bottleneck_blocks([getattr(r, "layer%d" % li) for li in range(1, 5)], dt, temporal=False)
        self.trunk = BottleneckTrunk(stem, r.bn1, ((1, 3, 3), (1, 2, 2), (0, 1, 1)), blocks)

    def conv_layers(self):
        return self.trunk.conv_layers()

    def flush_grads(self):
        self.trunk.flush_grads()

    def forward(self, x: torch.Tensor, mode: str, groups: int = 1):
        """x: (N,3,H,W) fp32 -> (embedding (N,128) unit-norm, tape). groups > 1 (train mode): x holds `groups` batches the reference passes through fb in
        separate calls (the two anonymised views, train_anonymizer.py:153-157): separate batch statistics per block, running statistics moved block by block."""
        assert mode in ("eval", "train")
        E.require_cuda(x, "FBTrainer")
        r, mlp = self.m[0], self.m[1]
        f, tape = self.trunk.forward(E.clip_to_act(x.unsqueeze(2), cpad=4, dtype=r.compute_dtype), mode == "train", groups=groups if mode == "train" else 1)
        tape["mode"], tape["x_shape"] = mode, tuple(x.shape)
        h = head.linear(f, mlp.fc1.weight, mlp.fc1.bias, relu=True)
        g = head.linear(h, mlp.fc2.weight, mlp.fc2.bias)
        tape["h"], tape["g"] = h, g
        return head.l2_normalize(g), tape

    def grad_buckets(self):
        return [list(self.m[1].parameters())] + self.trunk.stage_params()

    def backward(self, tape, demb: torch.Tensor, dx_out: Optional[torch.Tensor] = None, on_bucket_done=None):
        mlp = self.m[1]
        train = tape["mode"] == "train"
        f, h, g = tape["f"], tape["h"], tape["g"]
        dg = _l2norm_bwd(g, demb)
        dh, dw2, db2 = _linear_bwd(h, mlp.fc2.weight, dg)
        dz1 = _mul(dh, (h > 0).float(), 1.0)
        df, dw1, db1 = _linear_bwd(f, mlp.fc1.weight, dz1)
        if train:
            _acc_grad(mlp.fc2.weight, dw2); _acc_grad(mlp.fc2.bias, db2)
            _acc_grad(mlp.fc1.weight, dw1); _acc_grad(mlp.fc1.bias, db1)
        if train and on_bucket_done is not None:
            on_bucket_done(0)
        dimg = self.trunk.backward(tape, df, on_stage_done=None if on_bucket_done is None else (lambda k: on_bucket_done(k + 1)))
        if train:
            return None
        N, _, Hh, Ww = tape["x_shape"]
        dview = Act(dimg.buf.view(N, 1, Hh, Ww, 4), 4)
        if dx_out is None:
            dx_out = torch.empty(tape["x_shape"], dtype=torch.float32, device=dimg.buf.device)
        TE.act_to_nchw_into(dview, 3, dx_out.unsqueeze(2))
        return dx_out


# ------------------------------------------------------------------------------------------------------------------
# UNet (train mode)
# ------------------------------------------------------------------------------------------------------------------

class UNetTrainer:
    ENC = (64, 128, 256, 512)

    def __init__(self, unet):
        self.m = unet
        dt = unet.compute_dtype

        def dc(module):
            seq = module.double_conv
            return [(TE.ConvLayer(seq[i].weight, seq[i].bias, (1, 1, 1), (0, 1, 1), dtype=dt), seq[i + 1]) for i in (0, 3)]

        self.inc = dc(unet.inc)
        self.down = [dc(getattr(unet, "down%d" % i).maxpool_conv[1]) for i in (1, 2, 3, 4)]
        self.up = [dc(getattr(unet, "up%d" % i).conv) for i in (1, 2, 3, 4)]
        self.outc = TE.ConvLayer(unet.outc.conv.weight, unet.outc.conv.bias, (1, 1, 1), (0, 0, 0), dtype=dt)
        self.refresh = TE.WeightRefresh(self.conv_layers)

    def conv_layers(self):
        out = [c for c, _ in self.inc] + [self.outc]
        for units in self.down + self.up:
            out += [c for c, _ in units]
        return out

    def flush_grads(self):
        TE.flush_conv_grads(self.conv_layers())

    def forward(self, x: torch.Tensor, groups: int = 1):
        """x: (N,3,H,W) fp32 -> (y (N,3,H,W) fp32, tape). BatchNorm2d uses the batch statistics of this call
        and updates the running stats once (the reference calls fa on the B*48 pseudo-images at once, Q2/Q14).
        groups > 1: x holds `groups` batches of N / groups images that the reference passes through the module in SEPARATE calls (the two VISPR views,
        train_anonymizer.py:80-84): every BatchNorm keeps one set of batch statistics per block and moves its running statistics once per block, in order."""
        m = self.m
        E.require_cuda(x, "UNetTrainer")
        assert x.shape[0] % groups == 0
        self.refresh.run()
        n, _, H, W = x.shape
        tdt = E.DTYPES[m.compute_dtype][0]
        a = E.clip_to_act(x.unsqueeze(2), cpad=8, dtype=m.compute_dtype)
        tape = dict(n=n, H=H, W=W, enc=[], dec=[])
        cats, cur, h, w = [], a, H, W
        for lvl in range(4):
            units = self.inc if lvl == 0 else self.down[lvl - 1]
            rec = {}
            if lvl > 0:
                rec["pool_in"] = cur
                cur, rec["pool_idx"] = E.maxpool(cur, (1, 2, 2), (1, 2, 2), return_idx=True)
                h, w = h // 2, w // 2
            cat = Act.empty(n, 1, h, w, 2 * self.ENC[lvl], tdt, x.device)
            mid, rec["u1"] = TE.conv_bn_act_train(units[0][0], units[0][1], cur, groups=groups)
            skip, rec["u2"] = TE.conv_bn_act_train(units[1][0], units[1][1], mid, out=cat.slice(0, self.ENC[lvl]), groups=groups)
            cats.append(cat)
            tape["enc"].append(rec)
            cur = skip
        rec = dict(pool_in=cur)
        cur, rec["pool_idx"] = E.maxpool(cur, (1, 2, 2), (1, 2, 2), return_idx=True)
        mid, rec["u1"] = TE.conv_bn_act_train(self.down[3][0][0], self.down[3][0][1], cur, groups=groups)
        cur, rec["u2"] = TE.conv_bn_act_train(self.down[3][1][0], self.down[3][1][1], mid, groups=groups)
        tape["bottom"] = rec
        for i, lvl in zip((0, 1, 2, 3), (3, 2, 1, 0)):
            cat = cats[lvl]
            _, _, sh_, sw_ = cat.dims
            _, _, ch, cw = cur.dims
            dy, dx = sh_ - 2 * ch, sw_ - 2 * cw
            E.upsample2x_into(cur, cat.slice(self.ENC[lvl], self.ENC[lvl]), dy // 2, dx // 2)
            rec = dict(lvl=lvl, in_hw=(ch, cw), pad=(dy // 2, dx // 2))
            mid, rec["u1"] = TE.conv_bn_act_train(self.up[i][0][0], self.up[i][0][1], cat, groups=groups)
            cur, rec["u2"] = TE.conv_bn_act_train(self.up[i][1][0], self.up[i][1][1], mid, groups=groups)
            tape["dec"].append(rec)
        tape["u4"] = cur
        logits = self.outc.forward(cur, relu=False, sigmoid=True)       # 1x1 conv + bias + sigmoid fused
        y = E.act_to_nchw(logits, m.n_classes).squeeze(2)
        tape["y"] = y
        TE.flush_counters()
        return y, tape

    @staticmethod
    def _unit_params(units):
        out = []
        for c, bn in units:
            out += [c.weight] + ([c.bias] if c.bias is not None else []) + [bn.weight, bn.bias]
        return out

    def grad_buckets(self):
        """Parameters in the order `backward` finishes them: [outc + up4, up3, up2, up1, down4, down3, down2, down1, inc]."""
        b = [[self.outc.weight, self.outc.bias] + self._unit_params(self.up[3])]
        b += [self._unit_params(self.up[i]) for i in (2, 1, 0)]
        b += [self._unit_params(self.down[i]) for i in (3, 2, 1, 0)]
        b.append(self._unit_params(self.inc))
        return b

    def backward(self, tape, dy: torch.Tensor, on_bucket_done=None):
        """dy: (N,3,H,W) fp32 gradient w.r.t. the UNet output; accumulates every parameter's .grad.
        on_bucket_done(k): the k-th group of `grad_buckets()` is final and flushed (last backward pass of a step only)."""
        def done(k, units, extra=()):
            if on_bucket_done is not None:
                TE.flush_conv_grads([c for c, _ in units] + list(extra))
                on_bucket_done(k)

        n, H, W = tape["n"], tape["H"], tape["W"]
        dlogit = TE.nchw_grad_to_act(dy, tape["y"], (1, H, W), dtype=self.m.compute_dtype)     # sigmoid backward fused
        self.outc.wgrad(tape["u4"], dlogit)
        d = self.outc.dgrad(dlogit, tape["u4"].dims[1:])
        dskip = {}
        for j, rec in enumerate(reversed(tape["dec"])):           # up4, up3, up2, up1
            lvl = rec["lvl"]
            dmid, _ = TE.conv_bn_act_train_bwd(rec["u2"], d)
            dcat, _ = TE.conv_bn_act_train_bwd(rec["u1"], dmid)
            c = self.ENC[lvl]
            dskip[lvl] = dcat.slice(0, c)
            d = TE.upsample2x_bwd(dcat.slice(c, c), rec["in_hw"][0], rec["in_hw"][1], rec["pad"][0], rec["pad"][1])
            done(j, self.up[3 - j], extra=(self.outc,) if j == 0 else ())
        rec = tape["bottom"]
        dmid, _ = TE.conv_bn_act_train_bwd(rec["u2"], d)
        d, _ = TE.conv_bn_act_train_bwd(rec["u1"], dmid)
        d = TE.maxpool_bwd(rec["pool_in"], rec["pool_idx"], d, (1, 2, 2), (1, 2, 2), add=dskip[3])
        done(4, self.down[3])
        for lvl in (3, 2, 1, 0):
            rec = tape["enc"][lvl]
            dmid, _ = TE.conv_bn_act_train_bwd(rec["u2"], d)
            d, _ = TE.conv_bn_act_train_bwd(rec["u1"], dmid, need_dx=lvl > 0)
            if lvl > 0:
                d = TE.maxpool_bwd(rec["pool_in"], rec["pool_idx"], d, (1, 2, 2), (1, 2, 2), add=dskip[lvl - 1])
            done(8 - lvl, self.inc if lvl == 0 else self.down[lvl - 1])


# ------------------------------------------------------------------------------------------------------------------
# UNet++ (the reference's default anonymizer, train mode)
# ------------------------------------------------------------------------------------------------------------------

class UNetPPTrainer:
    """smp's UnetPlusPlus(resnet18, depth 4) of model_loaders.py:17-30 in train() mode (train_anonymizer.py:73-123 with the default
    `arch`): batch-statistics BatchNorm2d everywhere (running stats updated once per call), forward with a tape and the explicit
    backward of the dense skip pathway. The launch plan is unetpp.UnetPlusPlus.forward's: producers write into their channel slice
    of the consumer's concat buffer; backward, the gradient of a concat buffer is cut into the same slices, every tensor with several
    consumers (f1, f2, f3, x_1_1, x_2_2) sums its slices (tedspad_add_channels), the nearest x2 upsample sums its 2 x 2 blocks
    (tedspad_upsample_nearest2x_bwd). `encoder.layer4.*` is not on the path (depth 4) and receives no gradient, as in smp."""

    def __init__(self, m):
        self.m = m
        dt = m.compute_dtype
        enc = m.encoder
        CL = TE.ConvLayer
        self.stem, self.stem_bn = CL(enc.conv1.weight, None, (1, 2, 2), (0, 3, 3), pair_w=3, dtype=dt), enc.bn1
        self.blocks = []
        for li in (1, 2, 3):
            for bi, blk in enumerate(getattr(enc, "layer%d" % li)):
                s = blk.stride
                self.blocks.append(dict(
                    li=li, bi=bi, bn1=blk.bn1, bn2=blk.bn2, bnd=blk.downsample[1] if blk.downsample is not None else None,
                    c1=CL(blk.conv1.weight, None, (1, s, s), (0, 1, 1), dtype=dt), c2=CL(blk.conv2.weight, None, (1, 1, 1), (0, 1, 1), dtype=dt),
                    cd=CL(blk.downsample[0].weight, None, (1, s, s), (0, 0, 0), dtype=dt) if blk.downsample is not None else None))
        self.dec = {name: [(CL(b.conv1[0].weight, None, (1, 1, 1), (0, 1, 1), dtype=dt), b.conv1[1]),
                           (CL(b.conv2[0].weight, None, (1, 1, 1), (0, 1, 1), dtype=dt), b.conv2[1])] for name, b in m.decoder.blocks.items()}
        head_ = m.segmentation_head[0]
        self.head = CL(head_.weight, head_.bias, (1, 1, 1), (0, 1, 1), dtype=dt)
        self.refresh = TE.WeightRefresh(self.conv_layers)

    # the decoder blocks in the order `backward` finishes them
    DEC_ORDER = ("x_0_3", "x_0_2", "x_1_2", "x_0_1", "x_2_2", "x_1_1", "x_0_0")

    def conv_layers(self):
        out = [self.stem, self.head]
        for d in self.blocks:
            out += [d[k] for k in ("c1", "c2", "cd") if d[k] is not None]
        for units in self.dec.values():
            out += [c for c, _ in units]
        return out

    def flush_grads(self):
        TE.flush_conv_grads(self.conv_layers())

    def _stage_layers(self, li):
        return [d[k] for d in self.blocks if d["li"] == li for k in ("c1", "c2", "cd") if d[k] is not None]

    def off_path_params(self):
        """`encoder.layer4.*`: smp's ResNetEncoder keeps it at depth 4 but never runs it -- no gradient (torch leaves .grad None)."""
        return list(self.m.encoder.layer4.parameters())

    def grad_buckets(self):
        """Parameters in the order `backward` finishes them: [head + x_0_3, x_0_2, x_1_2, x_0_1, x_2_2, x_1_1, x_0_0, layer3, layer2, layer1, stem]."""
        b = []
        for i, name in enumerate(self.DEC_ORDER):
            b.append(([self.head.weight, self.head.bias] if i == 0 else []) + UNetTrainer._unit_params(self.dec[name]))
        for li in (3, 2, 1):
            g = []
            for d in self.blocks:
                if d["li"] == li:
                    for c, bn in ((d["c1"], d["bn1"]), (d["c2"], d["bn2"]), (d["cd"], d["bnd"])):
                        if c is not None:
                            g += [c.weight, bn.weight, bn.bias]
            b.append(g)
        b.append([self.stem.weight, self.stem_bn.weight, self.stem_bn.bias])
        return b

    def forward(self, x: torch.Tensor):
        """x: (N,3,H,W) fp32, H and W multiples of 16 -> (y (N,3,H,W) fp32 (no activation), tape)."""
        from .unetpp import UnetPlusPlus as U
        m = self.m
        E.require_cuda(x, "UNetPPTrainer")
        self.refresh.run()
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("expected (N,3,H,W), got %s" % (tuple(x.shape),))
        n, _, H, W = x.shape
        if H % 16 or W % 16:
            raise RuntimeError("Wrong input shape height=%d, width=%d. Expected image height and width divisible by 16." % (H, W))
        tdt = E.DTYPES[m.compute_dtype][0]

        def buf(c, div):
            return Act.empty(n, 1, H // div, W // div, c, tdt, x.device)

        B00, B11, B22 = buf(384, 8), buf(192, 4), buf(128, 2)
        B01, B12, B02 = buf(384, 4), buf(192, 2), buf(320, 2)
        tape = dict(n=n, H=H, W=W, enc=[], dec={})
        a = E.clip_to_act(x.unsqueeze(2), cpad=4, dtype=m.compute_dtype)
        f1, tape["stem"] = TE.conv_bn_act_train(self.stem, self.stem_bn, a, out=B22.slice(64, 64))
        cur, tape["idx1"] = E.maxpool(f1, (1, 3, 3), (1, 2, 2), pads=(0, 1, 1), return_idx=True)
        outs = {1: B11.slice(128, 64), 2: B00.slice(256, 128), 3: None}
        feats = {1: f1}
        for d in self.blocks:
            rec = {}
            h, rec["u1"] = TE.conv_bn_act_train(d["c1"], d["bn1"], cur)
            if d["cd"] is not None:
                r, rec["ud"] = TE.conv_bn_act_train(d["cd"], d["bnd"], cur, relu=False)
            else:
                r = cur
            cur, rec["u2"] = TE.conv_bn_act_train(d["c2"], d["bn2"], h, relu=True, residual=r, out=outs[d["li"]] if d["bi"] == 1 else None)
            tape["enc"].append(rec)
            if d["bi"] == 1:
                feats[d["li"] + 1] = cur
        f2, f3, f4 = feats[2], feats[3], feats[4]
        tape["f1"] = f1

        def block(name, cat, out=None):
            (c1, bn1), (c2, bn2) = self.dec[name]
            mid, u1 = TE.conv_bn_act_train(c1, bn1, cat)
            y, u2 = TE.conv_bn_act_train(c2, bn2, mid, out=out)
            tape["dec"][name] = (u1, u2)
            return y

        U._up_into(f4, B00.slice(0, 256))
        x00 = block("x_0_0", B00)
        U._up_into(f3, B11.slice(0, 128))
        x11 = block("x_1_1", B11, out=B01.slice(256, 64))
        U._up_into(f2, B22.slice(0, 64))
        x22 = block("x_2_2", B22, out=B12.slice(64, 64))
        U._copy_into(f2, B01.slice(320, 64))
        U._up_into(x00, B01.slice(0, 256))
        x01 = block("x_0_1", B01)
        U._copy_into(f1, B12.slice(128, 64))
        U._up_into(x11, B12.slice(0, 64))
        x12 = block("x_1_2", B12, out=B02.slice(128, 64))
        U._copy_into(Act(B12.buf, 128, B12.coff + 64), B02.slice(192, 128))
        U._up_into(x01, B02.slice(0, 128))
        x02 = block("x_0_2", B02)
        B03 = buf(64, 1)
        U._up_into(x02, B03)
        x03 = block("x_0_3", B03)
        tape["x03"] = x03
        y = self.head.forward(x03, relu=False)
        TE.flush_counters()
        return E.act_to_nchw(y, 3).squeeze(2), tape

    def backward(self, tape, dy: torch.Tensor, on_bucket_done=None):
        """dy: (N,3,H,W) fp32 gradient w.r.t. the output; accumulates every on-path parameter's .grad.
        on_bucket_done(k): the k-th group of `grad_buckets()` is final and flushed (last backward pass of a step only)."""
        n, H, W = tape["n"], tape["H"], tape["W"]
        code = _lib.F16 if self.m.compute_dtype == "f16" else _lib.BF16
        G = {}

        def acc(name, g: Act):
            if name not in G:
                G[name] = g
            else:
                t = G[name]
                nn_, _, h, w = g.dims
                check(_lib.lib().tedspad_add_channels(g.ptr, t.ptr, nn_ * h * w, g.c, g.ld, t.ld, code, _stream_ptr()), "tedspad_add_channels")

        def up_bwd(d: Act) -> Act:
            nn_, _, ho, wo = d.dims
            dx = Act.empty(nn_, 1, ho // 2, wo // 2, d.c, d.buf.dtype, d.buf.device)
            check(_lib.lib().tedspad_upsample_nearest2x_bwd(d.ptr, dx.ptr, nn_, ho // 2, wo // 2, d.c, d.ld, dx.ld, 0, code, _stream_ptr()),
                  "tedspad_upsample_nearest2x_bwd")
            return dx

        def block_bwd(name, d: Act) -> Act:
            u1, u2 = tape["dec"][name]
            dmid, _ = TE.conv_bn_act_train_bwd(u2, d)
            dcat, _ = TE.conv_bn_act_train_bwd(u1, dmid)
            return dcat

        def done(k, layers):
            if on_bucket_done is not None:
                TE.flush_conv_grads(layers)
                on_bucket_done(k)

        dl = TE.nchw_grad_to_act(dy, None, (1, H, W), dtype=self.m.compute_dtype)
        x03 = tape["x03"]
        self.head.wgrad(x03, dl)
        d = self.head.dgrad(dl, x03.dims[1:])
        dB = block_bwd("x_0_3", d)
        acc("x02", up_bwd(dB))
        done(0, [self.head] + [c for c, _ in self.dec["x_0_3"]])
        dB = block_bwd("x_0_2", G["x02"])                                   # [up(x01) | x12 | x22 | f1]
        acc("x01", up_bwd(dB.slice(0, 128))); acc("x12", dB.slice(128, 64)); acc("x22", dB.slice(192, 64)); acc("f1", dB.slice(256, 64))
        done(1, [c for c, _ in self.dec["x_0_2"]])
        dB = block_bwd("x_1_2", G["x12"])                                   # [up(x11) | x22 | f1]
        acc("x11", up_bwd(dB.slice(0, 64))); acc("x22", dB.slice(64, 64)); acc("f1", dB.slice(128, 64))
        done(2, [c for c, _ in self.dec["x_1_2"]])
        dB = block_bwd("x_0_1", G["x01"])                                   # [up(x00) | x11 | f2]
        acc("x00", up_bwd(dB.slice(0, 256))); acc("x11", dB.slice(256, 64)); acc("f2", dB.slice(320, 64))
        done(3, [c for c, _ in self.dec["x_0_1"]])
        dB = block_bwd("x_2_2", G["x22"])                                   # [up(f2) | f1]
        acc("f2", up_bwd(dB.slice(0, 64))); acc("f1", dB.slice(64, 64))
        done(4, [c for c, _ in self.dec["x_2_2"]])
        dB = block_bwd("x_1_1", G["x11"])                                   # [up(f3) | f2]
        acc("f3", up_bwd(dB.slice(0, 128))); acc("f2", dB.slice(128, 64))
        done(5, [c for c, _ in self.dec["x_1_1"]])
        dB = block_bwd("x_0_0", G["x00"])                                   # [up(f4) | f3]
        acc("f4", up_bwd(dB.slice(0, 256))); acc("f3", dB.slice(256, 128))
        done(6, [c for c, _ in self.dec["x_0_0"]])
        # ---- encoder: layer3 <- d(f4); its input gradient joins d(f3); ... -----------------------------------------------------------
        d = G["f4"]
        for i in range(len(self.blocks) - 1, -1, -1):
            blk, rec = self.blocks[i], tape["enc"][i]
            dh, dres = TE.conv_bn_act_train_bwd(rec["u2"], d)
            t = TE.conv_bn_act_train_bwd(rec["ud"], dres)[0] if "ud" in rec else dres
            d, _ = TE.conv_bn_act_train_bwd(rec["u1"], dh, dx_residual=t)
            if blk["bi"] == 0:
                done(7 + 3 - blk["li"], self._stage_layers(blk["li"]))
                if blk["li"] > 1:
                    acc("f%d" % blk["li"], d)
                    d = G["f%d" % blk["li"]]
        d = TE.maxpool_bwd(tape["f1"], tape["idx1"], d, (1, 3, 3), (1, 2, 2), pads=(0, 1, 1), add=G["f1"])
        TE.conv_bn_act_train_bwd(tape["stem"], d, need_dx=False)
        done(10, [self.stem])
