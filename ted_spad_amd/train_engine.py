"""Training-path plumbing on top of `engine`: a trainable convolution (forward with batch
statistics, data gradient, weight gradient), train-mode BatchNorm, and the backward launchers
of the pooling / resize ops. All arithmetic is in libtedspad_hip.so; torch holds the buffers.

Backward of the reference's torch.nn modules inside `loss.backward()`
(anonymization_training/train_anonymizer.py:122,190-191):

  * data gradient of a convolution = a convolution of dY with the spatially flipped, channel-
    transposed weights; a stride-s conv splits into s^d dense sub-convolutions, one per parity class
    of the input position, each writing its results interleaved in place (tedspad_conv_extras.out_*).
    It runs on the SAME implicit-GEMM kernel as the forward pass.
  * the ReLU backward is folded into the data-gradient conv that produces d(input): the input IS
    the ReLU output, so the conv's epilogue masks with it (tedspad_conv_extras.mask).
  * eval-mode BatchNorm backward (phase 1: frozen ft, train_anonymizer.py:73-75) is the folded
    scale, multiplied into the data-gradient weights.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
from typing import List, Optional, Tuple

import torch

from . import _lib, engine as E
from ._lib import PoolDesc, check
from .engine import Act, PackedConv, _stream_ptr


class ZeroArena:
    """One big pre-zeroed fp32 buffer per training step for every atomic-accumulation target (BatchNorm batch
    statistics, per-channel gradient sums, packed weight-gradient matrices): ONE memset per step instead of one
    torch.zeros launch per request (~1500 per step). The request sequence repeats every step, so slices are
    handed out bump-style; `reset()` re-zeroes what the previous step used."""

    def __init__(self):
        self.buf = None
        self.used = 0
        self.high = 0
        self.gen = 0

    def reset(self, device):
        flush_deferred()                                        # pending gradients are views of the buffer zeroed below
        if self.buf is None or self.buf.device != torch.device(device):
            self.buf = torch.zeros(160 << 20, dtype=torch.float32, device=device)    # 640 MB (all three nets' packed weight gradients fit)
        elif self.high:
            self.buf[: self.high].zero_()
        self.used, self.high = 0, 0
        self.gen += 1

    def take(self, shape, device):
        n = 1
        for d in shape:
            n *= int(d)
        n4 = (n + 3) // 4 * 4                                   # keep 16-byte alignment
        if NO_ARENA:
            return torch.zeros(tuple(shape), dtype=torch.float32, device=device)
        if self.buf is None or self.buf.device != torch.device(device):
            self.reset(device)
        if self.used + n4 > self.buf.numel():                   # outgrown: fresh zeros for this request
            return torch.zeros(tuple(shape), dtype=torch.float32, device=device)
        out = self.buf[self.used: self.used + n].view(tuple(shape))
        self.used += n4
        self.high = max(self.high, self.used)
        return out


ARENA = ZeroArena()

# Tiny per-BatchNorm updates are collected and applied with multi-tensor launches: `num_batches_tracked += 1`
# (one launch per BN per forward otherwise) and d(gamma) / d(beta), whose per-call sums live in the arena (one clone
# or add launch per tensor per clip otherwise).
# the conv output in front of a train-mode BatchNorm (re-read by the BN apply and twice by the backward) kept in the 16-bit activation dtype, as
# the reference's autocast region holds it (train_anonymizer.py:78,151), instead of fp32: 2 of the 6-8 bytes per element each of those passes moves
TRAIN_Z16 = os.environ.get("TEDSPAD_TRAIN_Z16", "1") != "0"
# Gradients entering the networks through the autograd bridge (autograd.py: the reference's own training loop) are multiplied by this power of two and every
# gradient leaving them is divided by it again: gfx950's matrix cores flush f16 SUBNORMAL operands, and the per-pixel activation gradients of the anonymizer's
# high-resolution levels are below 6e-5 at the reference's scale. 1: the reference's literal arithmetic. (AnonymizerTrainStep has its own
# `loss_scale`, with the non-finite check of a GradScaler; the bridge cannot skip a step, an overflow shows as non-finite gradients exactly as in the reference's fp16 run: every f16 store of the training path -- conv epilogues
# with tedspad_conv_extras.nosat, csrc/train_ops.hip -- converts WITHOUT the +-65504 clamp the inference path has, so inf / NaN reach the parameter gradients and a
# torch GradScaler around the bridge backs off as it does in train_anonymized_action.py:92-94.)
GRAD_SCALE = float(os.environ.get("TEDSPAD_GRAD_SCALE", "256"))
DB_SLOTS = 64           # rows of the conv-bias gradient accumulator of tedspad_bn_bwd_apply
NO_ARENA = bool(os.environ.get("TEDSPAD_NO_ARENA"))       # every request its own torch.zeros (debugging)
REFRESH_IN_PLACE = os.environ.get("TEDSPAD_WEIGHT_REFRESH", "1") != "0"     # 0: every stale image rebuilt by the lazy path (A/B timing)
IMAGES_GEN = 0          # bumped whenever a ConvLayer builds a NEW kernel-form image (WeightRefresh then rebuilds its job tables)
_PENDING_COUNT = {}     # id(tensor) -> [tensor, increments]
_PENDING_GRAD = {}      # id(param)  -> [param, [arena slices]]


def bump_counter(t: torch.Tensor):
    e = _PENDING_COUNT.setdefault(id(t), [t, 0])
    e[1] += 1


def defer_grad(p, g: torch.Tensor):
    _PENDING_GRAD.setdefault(id(p), [p, []])[1].append(g)


def flush_counters():
    if _PENDING_COUNT:
        ent = list(_PENDING_COUNT.values())
        _PENDING_COUNT.clear()
        torch._foreach_add_([e[0] for e in ent], [e[1] for e in ent])


_SIDE = {}              # device -> [stream, busy]
WGRAD_STREAM = os.environ.get("TEDSPAD_WGRAD_STREAM", "1") != "0"


def side_stream(device):
    """The stream the weight-gradient kernels run on, or None: while the tile tuner is timing launches (a candidate measured with another
    stream's kernels on the chip can lose to a slower one), or with TEDSPAD_WGRAD_STREAM=0."""
    if not WGRAD_STREAM or E.tuning_pending() or E.DETERMINISTIC:           # (deterministic mode: one launch of a kernel family at a time, csrc/det_gate.h)
        return None                                     # (inside a stream capture the side stream joins the capture: a parallel branch of the graph)
    ent = _SIDE.get(device)
    if ent is None:
        ent = _SIDE[device] = [torch.cuda.Stream(device=device), False]       # (priority range here is (0, -1): 0 is already the lowest)
    ent[1] = True
    return ent[0]


def join_side_stream():
    """The current stream waits for the weight-gradient kernels launched so far (before their accumulators are read, or the arena is reset)."""
    for ent in _SIDE.values():
        if ent[1]:
            torch.cuda.current_stream().wait_stream(ent[0])
            ent[1] = False


def flush_deferred():
    """Apply the collected counter increments and BN parameter gradients (must run before the arena is reset)."""
    join_side_stream()
    flush_counters()
    if not _PENDING_GRAD:
        return
    ent = list(_PENDING_GRAD.values())
    _PENDING_GRAD.clear()
    fresh = [e for e in ent if e[0].grad is None]
    if fresh:                                               # .grad = copy of the first slice (one multi-tensor launch)
        for e, g in zip(fresh, torch._foreach_mul([e[1][0] for e in fresh], 1.0)):
            e[0].grad = g
            e[1] = e[1][1:]
    r = 0
    while True:
        todo = [e for e in ent if len(e[1]) > r]
        if not todo:
            break
        torch._foreach_add_([e[0].grad for e in todo], [e[1][r] for e in todo])
        r += 1


def _code(t: torch.Tensor) -> int:
    return _lib.F16 if t.dtype == torch.float16 else _lib.BF16


def _ceil_div(a, b):
    return -(-a // b)


class DgradPlan:
    """The data gradient of one convolution as dense sub-convolutions over dY."""

    def __init__(self, w5: torch.Tensor, scale: Optional[torch.Tensor], stride_k, pads_front_k, in_dims, out_dims, dtype, pair_w=None):
        """w5: the fp32 (co,ci,kt,kh,kw) parameter; stride_k / pads_front_k / in_dims describe the conv in KERNEL form
        (the stem: pixel-pair form, stride (2,2,1))."""
        co, ci, kt, kh, kw = w5.shape
        if pair_w is not None:
            pw2 = (pair_w + 1) // 2
            kw = (kw + 2 * pw2 - pair_w + 1) // 2
            cin_k = 8
        else:
            cin_k = (ci + 7) // 8 * 8
        self.ci = cin_k
        self.in_dims, self.out_dims = tuple(in_dims), tuple(out_dims)
        self.subs: List[Tuple] = []
        self.need_zero = False
        ks = (kt, kh, kw)
        wscale = None if scale is None else scale.detach().float().contiguous()
        for rt in range(stride_k[0]):
            for rh in range(stride_k[1]):
                for rw in range(stride_k[2]):
                    r = (rt, rh, rw)
                    E, cs, pf2, J = [], [], [], []
                    empty = False
                    for dim in range(3):
                        s, k, pf, n_in, n_out = stride_k[dim], ks[dim], pads_front_k[dim], in_dims[dim], out_dims[dim]
                        c, q = (r[dim] + pf) % s, (r[dim] + pf) // s
                        e_n = _ceil_div(k - c, s) if k > c else 0
                        j_n = _ceil_div(n_in - r[dim], s) if n_in > r[dim] else 0
                        if e_n == 0 or j_n == 0:
                            empty = True
                            break
                        p2 = e_n - 1 - q
                        assert p2 >= 0, "unsupported padding/stride combination in dgrad"
                        if j_n > n_out + p2:      # trailing input the forward conv never read
                            j_n = n_out + p2
                            self.need_zero = True
                        E.append(e_n); cs.append(c); pf2.append(p2); J.append(j_n)
                    if empty:
                        self.need_zero = True
                        continue
                    geo = tuple(E) + tuple(cs) + tuple(stride_k)            # tap = c + s*(E-1-e): the flipped taps of this class
                    pc = PackedConv.dgrad_sub(w5, wscale, geo, pair_w, dtype)
                    pc.nosat = True        # training path: f16 stores do not saturate (an overflow must reach the non-finite check)
                    self.subs.append((r, pc, tuple(pf2), tuple(J)))
        self.stride = tuple(stride_k)

    def run(self, dy: Act, residual: Optional[Act] = None, mask: Optional[Act] = None, out: Optional[Act] = None) -> Act:
        n = dy.dims[0]
        assert dy.dims[1:] == self.out_dims, (dy.dims, self.out_dims)
        if out is None:
            out = Act.empty(n, *self.in_dims, self.ci, dy.buf.dtype, dy.buf.device)
        if self.need_zero:
            if residual is not None:
                raise NotImplementedError("dgrad with a fused residual needs every input position covered by the conv "
                                          "(not the case for this stride/kernel); add the residual separately")
            out.buf.zero_()
        dense = self.stride == (1, 1, 1)
        for r, pc, pf2, J in self.subs:
            pc(dy, pads=pf2, out=out, out_dims=J, residual=residual, mask=mask, relu=False,
               out_map=None if dense and J == self.in_dims else (self.stride, r))
        return out


class ConvLayer:
    """A trainable convolution bound to its fp32 parameter(s) (reference layout (co,ci,*k)).
    Repacks the 16-bit kernel-layout copies whenever the parameter was updated."""

    def __init__(self, weight, bias=None, stride=(1, 1, 1), pads=(0, 0, 0), pads_back=None, pair_w=None, dtype=E.DEFAULT_DTYPE):
        self.weight, self.bias = weight, bias
        self.stride, self.pads = tuple(stride), tuple(pads)
        self.pads_back = self.pads if pads_back is None else tuple(pads_back)
        self.pair_w, self.dtype = pair_w, dtype
        self._fwd = {}            # flavour (scale is None, shift is None) -> [sig, PackedConv, scale, shift]
        self._last = None         # the forward image used last (geometry queries)
        self._dgrad = {}          # (x dims, dy dims, scale is None) -> [sig, DgradPlan, scale]

    # ---- kernel-form weights ------------------------------------------------------------------------
    def _w5(self):
        w = self.weight.detach()
        return w.unsqueeze(2) if w.dim() == 4 else w

    def _sig(self, scale, shift):
        def v(t):       # _tedspad_rev: in-place re-folds of a BatchNorm vector (cached_fold / WeightRefresh), which torch's version counter does not see
            return None if t is None else (t.data_ptr(), t._version, getattr(t, "_tedspad_rev", 0))
        return (v(self.weight), v(self.bias), v(scale), v(shift))

    def fwd_conv(self, scale=None, shift=None) -> PackedConv:
        """scale/shift None -> (1, bias): the raw conv of the train-mode path. One image per flavour (train mode / folded BatchNorm):
        a network that alternates between them (ft: frozen in phase 1, trained in phase 2) keeps both, and `WeightRefresh` rewrites them
        in place after an optimizer step; this lazy path only builds what does not exist yet (or was built for other tensors)."""
        global IMAGES_GEN
        key = (scale is None, shift is None)
        sig = self._sig(scale, shift)
        ent = self._fwd.get(key)
        if ent is None or ent[0] != sig:
            w = self._w5()
            sf = (self.bias.detach() if self.bias is not None else None) if shift is None else shift
            pc = PackedConv(w, scale, sf, stride=self.stride, dtype=self.dtype, device=w.device, pair_w=self.pair_w)
            pc.nosat = True
            old = ent[1] if ent is not None else self._last
            if old is not None:   # same geometry: keep the gather tables and the tuned tile choice
                pc._ktabs, pc._cfgs = old._ktabs, old._cfgs
            ent = [sig, pc, scale, shift]
            self._fwd[key] = ent
            IMAGES_GEN += 1
        self._last = ent[1]
        return ent[1]

    def geom_conv(self) -> PackedConv:
        """Any forward image of this conv (kernel-form geometry, gather tables, K padding: the same for every flavour)."""
        return self._last if self._last is not None else self.fwd_conv()

    def _pads_k(self, pc: PackedConv):
        if self.pair_w is None:
            return self.pads, self.pads_back
        return (self.pads[0], self.pads[1], pc.pair_pw), (self.pads_back[0], self.pads_back[1], pc.k[2] - 1 - pc.pair_pw)

    def forward(self, x: Act, scale=None, shift=None, relu=False, residual=None, stats=None, out=None, sigmoid=False, y32=False):
        pc = self.fwd_conv(scale, shift)
        pk, pbk = self._pads_k(pc)
        return pc(x, pads=pk, pads_back=pbk, relu=relu, residual=residual, stats=stats, out=out, sigmoid=sigmoid, y32=y32)

    # ---- backward -----------------------------------------------------------------------------------
    def dgrad(self, dy: Act, x_dims, scale=None, residual=None, mask=None, out=None) -> Act:
        """d(input) (kernel-form channels: the stem returns the (n,t,h,w/2,8) pixel-pair tensor == (n,t,h,w,4))."""
        global IMAGES_GEN
        key = (tuple(x_dims), dy.dims[1:], scale is None)
        sig = self._sig(scale, None)
        plan = self._dgrad.get(key)
        if plan is None or plan[0] != sig:
            pc = self.geom_conv()
            p_k, _ = self._pads_k(pc)
            new = DgradPlan(self._w5().float(), scale, pc.stride, p_k, x_dims, dy.dims[1:], self.dtype, pair_w=self.pair_w)
            if plan is not None:
                for (_, pc_new, _, _), (_, pc_old, _, _) in zip(new.subs, plan[1].subs):
                    pc_new._ktabs, pc_new._cfgs = pc_old._ktabs, pc_old._cfgs
            plan = [sig, new, scale]
            self._dgrad[key] = plan
            IMAGES_GEN += 1
        return plan[1].run(dy, residual=residual, mask=mask, out=out)

    def wgrad(self, x: Act, dy: Act, db: Optional[torch.Tensor] = None):
        """Accumulates d(weight) in the packed [cout_pad][kpad] fp32 layout (float atomics); several calls per step
        (the three clips) add into the same matrix. `flush_grad()` converts it to the parameter layout once."""
        pc = self.geom_conv()
        n, t, h, w = x.dims
        pk, _ = self._pads_k(pc)
        if getattr(self, "_dwp_gen", -1) != ARENA.gen or self._dwp is None:     # new step, or flushed since (one flush per backward pass in the autograd path)
            self._dwp = ARENA.take((pc.cpad, pc.kpad), x.buf.device)
            self._dwp_gen = ARENA.gen
            self._db = None
        to, ho, wo = dy.dims[1:]
        nc = n
        if n * max(t * h * w * x.ld, to * ho * wo * dy.ld) >= E.MAX_ELEMS or n * to * ho * wo >= E.MAX_WGRAD_PIXELS:
            nc = min(E.batch_chunk(n, [t * h * w * x.ld, to * ho * wo * dy.ld], E.MAX_ELEMS), E.batch_chunk(n, [to * ho * wo], E.MAX_WGRAD_PIXELS))
        # The weight gradient is off the backward pass's critical path (nothing reads it before the flush): it goes to a side stream, where
        # this L2->LDS-feed-bound kernel runs beside the HBM-bound BatchNorm passes and the data-gradient convs of the layers in front.
        side = side_stream(x.buf.device)
        launches = []
        for n0 in range(0, n, nc):                      # chunks of whole samples accumulate into the same matrix
            n1 = min(n, n0 + nc)
            xs, ds = (Act(x.buf[n0:n1], x.c, x.coff), Act(dy.buf[n0:n1], dy.c, dy.coff)) if nc < n else (x, dy)
            d = pc._desc(n1 - n0, t, h, w, xs.ld, pk, dy.dims[1:], ds.ld, 0, False)
            launches.append((d, xs, ds, pc._ktab(d)))
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())          # x, dy and the zeroed accumulator are ready
            if not torch.cuda.is_current_stream_capturing():       # (a graph's pool keeps its tensors alive by itself)
                x.buf.record_stream(side); dy.buf.record_stream(side)
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            for d, xs, ds, ktab in launches:
                check(_lib.lib().tedspad_conv_wgrad(C.byref(d), xs.ptr, ds.ptr, ktab.data_ptr(), self._dwp.data_ptr(), _stream_ptr()), "tedspad_conv_wgrad")
        if self.bias is not None:
            if db is None:                                     # (tedspad_bn_bwd_apply gathers it while it writes dy when a BatchNorm follows the conv)
                db = channel_sums(dy)[0]
            self._db = db if self._db is None else self._db + db

    def flush_grad(self):
        """d(weight) / d(bias) of this step -> `.grad` in the parameter layout (no-op if wgrad was not called)."""
        flush_deferred()
        if getattr(self, "_dwp_gen", -1) != ARENA.gen or self._dwp is None:
            return
        pc = self.geom_conv()
        w5 = self._w5()
        co, ci, kt, kh, kw = w5.shape
        kt_, kh_, kw_ = pc.k
        g = self._dwp[:co, :pc.K].view(co, kt_, kh_, kw_, pc.cin).permute(0, 4, 1, 2, 3)
        g = E.stem_pair_grad(g, ci, kw, self.pair_w) if self.pair_w is not None else g[:, :ci]
        g = g.reshape(self.weight.shape)
        rs = getattr(self, "_grad_row_scale", None)      # frozen-BN mode: d(conv output) = delta * folded BN scale per output channel
        if rs is not None:
            g = g * rs[:co].view([-1] + [1] * (g.dim() - 1))
            self._grad_row_scale = None
        if self.weight.grad is None:
            g = g.contiguous()
            if g.untyped_storage().data_ptr() == self._dwp.untyped_storage().data_ptr():
                g = g.clone()        # 1x1x1 convs: the view IS the arena slice, which the next step's reset zeroes
            self.weight.grad = g
        else:
            self.weight.grad.add_(g)                 # in place: .grad may be a view into a gradient bucket (grad_reduce.py)
        if self.bias is not None and self._db is not None:
            db = self._db[:co]
            if self.bias.grad is None:
                self.bias.grad = db.clone()
            else:
                self.bias.grad.add_(db)
        self._dwp = None


def flush_conv_grads(layers):
    """`flush_grad()` of every layer in ONE launch (tedspad_wgrad_unpack_multi): the packed accumulators of this step -> the parameters'
    .grad (accumulated when .grad exists -- a view into a gradient bucket, grad_reduce.py -- else created). The stem's pixel-pair form
    keeps the per-layer path; bias gradients are added with one multi-tensor launch. The job table is cached while the accumulators and gradients keep their addresses
    (the arena hands out the same slices every step)."""
    flush_deferred()
    jobs, keep, key, slow, badd = [], [], [], [], ([], [])
    for L in layers:
        if getattr(L, "_dwp_gen", -1) != ARENA.gen or L._dwp is None:
            continue
        if L.pair_w is not None or not L.weight.is_contiguous():
            slow.append(L)
            continue
        pc = L.geom_conv()
        co, ci, kt, kh, kw = L._w5().shape
        rs = getattr(L, "_grad_row_scale", None)
        acc = L.weight.grad is not None
        if acc and (not L.weight.grad.is_contiguous() or L.weight.grad.dtype != torch.float32):
            slow.append(L)
            continue
        if not acc:
            L.weight.grad = torch.empty_like(L.weight)
        g = L.weight.grad
        jobs.append(_lib.WgradUnpackJob(dw=L._dwp.data_ptr(), grad=g.data_ptr(), row_scale=rs.data_ptr() if rs is not None else None,
                                        co=co, ci=ci, kt=kt, kh=kh, kw=kw, cink=pc.cin, kpad=pc.kpad, accumulate=int(acc), block0=0, nblocks=0))
        keep += [L._dwp, g, rs]
        key.append((L._dwp.data_ptr(), g.data_ptr(), rs.data_ptr() if rs is not None else 0, acc, id(L)))
        L._grad_row_scale = None
        L._dwp = None
        if L.bias is not None and L._db is not None:
            db = L._db[:co]
            if L.bias.grad is None:
                L.bias.grad = db.clone()
            else:
                badd[0].append(L.bias.grad)
                badd[1].append(db)
    if badd[0]:
        torch._foreach_add_(badd[0], badd[1])
    if jobs:
        key = tuple(key)
        tab = _UNPACK_TABLES.get(key)
        if tab is None:
            if len(_UNPACK_TABLES) > 64:
                _UNPACK_TABLES.clear()
            tab = _UNPACK_TABLES[key] = E.JobTable(jobs)
        tab.keep = keep                     # this step's tensors (same addresses as the cached table's)
        tab.launch(keep[0].device)
    for L in slow:
        L.flush_grad()


_UNPACK_TABLES = {}


class WeightRefresh:
    """In-place refresh of every kernel-form image a set of ConvLayers holds (forward images of both flavours, data-gradient images) and
    of the folded BatchNorm vectors they were built with, after an optimizer step: ONE tedspad_fold_multi + ONE tedspad_pack_multi launch
    from static job tables instead of a rebuild (new buffers, one launch each, ~100 us of host time per image) by the lazy path. `folds`:
    the owner's fold cache {id(bn): [sig, scale, shift, bn, conv_bias]}. The lazy path stays responsible for images that do not exist yet;
    whenever it builds one (IMAGES_GEN moves) the tables are rebuilt."""

    def __init__(self, layers_fn, folds=None):
        self.layers_fn, self.folds = layers_fn, folds if folds is not None else {}
        self.gen, self.nfolds, self.fold_tab, self.pack_tab = -1, -1, None, None

    def _build(self, layers):
        fj, keep = [], []
        for ent in self.folds.values():
            _, s, b, bn, cb = ent
            fj.append(E.fold_job(bn, cb, s, b))
            keep += [s, b]
        pj = []
        for L in layers:
            w = L._w5()
            if not (w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()):
                raise _lib.TedSpadHipError("WeightRefresh: a conv parameter is not a contiguous fp32 CUDA tensor")
            for ent in L._fwd.values():
                _, pc, scale, shift = ent
                pj.append(pc.pack_job(w, None))
                if shift is None and L.bias is not None:          # train flavour of a conv with bias: the image's shift vector is a padded copy
                    fj.append(_lib.FoldJob(gamma=None, beta=None, mean=None, var=None, conv_bias=L.bias.data_ptr(), scale=None, shift=pc.shift.data_ptr(),
                                           scale2=None, shift2=None, eps=0.0, C=L.bias.numel(), n=pc.shift.numel(), n2=0, reserved=0))
                elif scale is not None and (getattr(scale, "_tedspad_padded", 0) < pc.cpad or (shift is not None and getattr(shift, "_tedspad_padded", 0) < pc.cpad)):
                    raise _lib.TedSpadHipError("WeightRefresh: a folded image whose scale / shift are private copies (not engine.fold_bn outputs)")
                keep += [pc, scale, shift]
            for ent in L._dgrad.values():
                _, plan, scale = ent
                for _, pc, _, _ in plan.subs:
                    pj.append(pc.pack_job(w, scale))
                    keep.append(pc)
                keep.append(scale)
        self.fold_tab, self.pack_tab = E.JobTable(fj, keep), E.JobTable(pj)
        self.gen, self.nfolds = IMAGES_GEN, len(self.folds)

    @staticmethod
    def _ptrs(sig):
        return tuple(None if v is None else v[0] for v in sig)

    def _changes(self, layers):
        """(anything stale, any tensor MOVED): a signature holds (data_ptr, version[, rev]) per tensor; a changed data_ptr (`p.data = ...`, `.to()`, a dtype cast,
        an EMA swap) means the job tables' source addresses are dead, not just that the values changed."""
        stale = moved = False
        for L in layers:
            for ent, new in [(e, L._sig(e[2], e[3])) for e in L._fwd.values()] + [(e, L._sig(e[2], None)) for e in L._dgrad.values()]:
                if ent[0] != new:
                    stale = True
                    moved = moved or self._ptrs(ent[0]) != self._ptrs(new)
        for ent in self.folds.values():
            new = fold_sig(ent[3], ent[4])
            if ent[0] != new:
                stale = True
                moved = moved or self._ptrs(ent[0]) != self._ptrs(new)
        return stale, moved

    def stale(self, layers) -> bool:
        return self._changes(layers)[0]

    def run(self):
        if not REFRESH_IN_PLACE:
            return False
        layers = self.layers_fn()
        if not layers:
            return False
        stale, moved = self._changes(layers)
        if not stale:
            return False
        if moved or self.gen != IMAGES_GEN or self.nfolds != len(self.folds) or self.pack_tab is None:
            self._build(layers)                 # (a moved parameter / BatchNorm tensor: the cached tables point at its old storage)
        dev = layers[0].weight.device
        self.fold_tab.launch(dev)
        self.pack_tab.launch(dev)
        for L in layers:
            for ent in L._fwd.values():
                ent[0] = L._sig(ent[2], ent[3])
            for ent in L._dgrad.values():
                ent[0] = L._sig(ent[2], None)
        for ent in self.folds.values():
            ent[0] = fold_sig(ent[3], ent[4])
        return True


def mark_updated(params):
    """Tell the weight images / BatchNorm folds built from `params` that the parameters changed. Needed after an optimizer whose step() writes the parameters WITHOUT
    bumping their version counters -- torch's `fused=True` implementations do (measured: `_version` stays 0 across `torch.optim.Adam(fused=True).step()`) -- since
    `WeightRefresh` recognises a change by (data_ptr, version, this revision). AnonymizerTrainStep calls it after its own fused steps; a caller that drives the
    modules through autograd.py with a fused optimizer must too."""
    for p in params:
        p._tedspad_rev = getattr(p, "_tedspad_rev", 0) + 1


def fold_sig(bn, conv_bias=None):
    """(data_ptr, version) of every tensor a fold reads. The running statistics are written by tedspad_bn_train_apply through raw pointers, which torch's version
    counter does not see: `num_batches_tracked` (bumped once per train-mode forward, train_engine.bump_counter) stands in for them, so a frozen-flavour forward after
    a train-mode one without an optimizer step (a skipped step under loss scaling, a no_grad train() pass) re-folds."""
    sig = tuple((t.data_ptr(), t._version + int(getattr(t, "_tedspad_rev", 0))) for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var) + ((conv_bias,) if conv_bias is not None else ()))
    nbt = getattr(bn, "num_batches_tracked", None)
    return sig + ((nbt.data_ptr(), nbt._version + int(getattr(nbt, "_tedspad_rev", 0))),) if nbt is not None else sig


def cached_fold(folds: dict, bn, conv_bias=None):
    """Eval-mode BatchNorm (+ conv bias) as zero-padded fp32 (scale, shift) from the cache `folds` ({id(bn): [sig, scale, shift, bn, conv_bias]}):
    folded once; re-folded INTO THE SAME TENSORS when the BatchNorm changed (the images built from them alias these vectors)."""
    sig = fold_sig(bn, conv_bias)
    hit = folds.get(id(bn))
    if hit is None:
        s, b = E.fold_bn(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, conv_bias=conv_bias)
        hit = folds[id(bn)] = [sig, s, b, bn, conv_bias]
    elif hit[0] != sig:
        E.JobTable([E.fold_job(bn, conv_bias, hit[1], hit[2])]).launch(hit[1].device)
        hit[0] = sig
        for t in hit[1:3]:
            t._tedspad_rev = getattr(t, "_tedspad_rev", 0) + 1       # images with the old scale folded in are stale
    return hit[1], hit[2]


# ---- per-channel reductions / BatchNorm ---------------------------------------------------------------

def channel_sums(dy: Act, y: Optional[Act] = None, z: Optional[torch.Tensor] = None, mean=None, invstd=None, relu=False, groups: int = 1,
                 gamma=None, beta=None, zcode=_lib.F32) -> torch.Tensor:
    """(2, C) fp32: row 0 = sum g, row 1 = sum g * xhat (zeros when z is None); g = dy * (y > 0 if relu).
    z: the fp32 (n,t,h,w,C) pre-normalisation conv output. groups > 1: (groups, 2, C), one set per block of n / groups samples."""
    n, t, h, w = dy.dims
    sums = ARENA.take((2, dy.c) if groups == 1 else (groups, 2, dy.c), dy.buf.device)
    check(_lib.lib().tedspad_bn_bwd_reduce(dy.ptr, y.ptr if y is not None else None, z.data_ptr() if z is not None else None, zcode,
                                           mean.data_ptr() if mean is not None else None, invstd.data_ptr() if invstd is not None else None,
                                           gamma.data_ptr() if gamma is not None else None, beta.data_ptr() if beta is not None else None, sums.data_ptr(), dy.c, n * t * h * w // groups, dy.c, dy.ld, y.ld if y is not None else 0,
                                           z.shape[-1] if z is not None else 0, int(relu), groups, _code(dy.buf), _stream_ptr()), "tedspad_bn_bwd_reduce")
    return sums


class BNTrainCtx:
    __slots__ = ("x", "z", "y", "mean", "invstd", "bn", "conv", "relu", "has_res", "groups", "zcode")


def conv_bn_act_train(conv: ConvLayer, bn, x: Act, relu=True, residual: Optional[Act] = None, out: Optional[Act] = None, groups: int = 1):
    """conv -> BatchNorm(batch statistics, running stats updated) -> (+residual) -> ReLU. Returns (y, ctx).
    The pre-normalisation conv output z (re-read by the BN apply and twice by the backward) is kept in the 16-bit activation dtype
    (TRAIN_Z16, the reference's autocast behaviour) or in fp32 (TEDSPAD_TRAIN_Z16=0); the batch sums come from the fp32 accumulators.
    groups > 1: the batch is `groups` consecutive blocks of samples, each normalised with its own batch statistics and the running
    statistics updated once per block, in order -- `groups` separate forward calls of the module (the three clips of a training step,
    train_anonymizer.py:169-175) as ONE launch sequence."""
    pc = conv.fwd_conv()
    stats = ARENA.take((2, pc.cpad) if groups == 1 else (groups, 2, pc.cpad), x.buf.device)
    if TRAIN_Z16:
        za = conv.forward(x, stats=stats)                            # Act, 16-bit; the batch sums in `stats` come from the fp32 accumulators
        z, zcode = za.buf, _code(za.buf)
        assert za.coff == 0 and za.ld == za.c
    else:
        z, zcode = conv.forward(x, stats=stats, y32=True), _lib.F32  # (n,t,h,w,cout) fp32
    n, t, h, w, cz = z.shape
    c = bn.weight.shape[0]
    assert n % groups == 0
    mean, invstd = ARENA.take((2, groups, cz), x.buf.device).unbind(0)
    for _ in range(groups):
        bump_counter(bn.num_batches_tracked)
    tdt = E.DTYPES[conv.dtype][0]
    y = out if out is not None else Act.empty(n, t, h, w, cz, tdt, z.device)
    # batch mean / variance -> scale / shift, the running-statistics update and the normalisation itself in ONE launch
    rows = n * t * h * w // groups
    check(_lib.lib().tedspad_bn_train_apply(z.data_ptr(), zcode, stats.data_ptr(), pc.cpad, rows, bn.weight.data_ptr(), bn.bias.data_ptr(),
                                            C.c_float(bn.eps), C.c_float(bn.momentum), bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                                            mean.data_ptr(), invstd.data_ptr(), c, residual.ptr if residual is not None else None, y.ptr,
                                            rows, cz, cz, residual.ld if residual is not None else 0, y.ld, int(relu), groups, _code(y.buf),
                                            _stream_ptr()), "tedspad_bn_train_apply")
    ctx = BNTrainCtx()
    ctx.x, ctx.z, ctx.y, ctx.mean, ctx.invstd, ctx.bn, ctx.conv, ctx.relu, ctx.has_res = x, z, y, mean, invstd, bn, conv, relu, residual is not None
    ctx.groups, ctx.zcode = groups, zcode
    return y, ctx


def conv_bn_act_train_bwd(ctx: BNTrainCtx, dy: Act, need_dx=True, dx_residual: Optional[Act] = None, dx_mask: Optional[Act] = None,
                          x_dims=None):
    """Backward of `conv_bn_act_train`: accumulates d(gamma), d(beta), d(weight), d(bias) into .grad and returns
    (dx or None, dres or None). dx = dgrad(dz) (+ dx_residual) masked by dx_mask."""
    bn, z, y = ctx.bn, ctx.z, ctx.y
    n, t, h, w, cz = z.shape
    c = bn.weight.shape[0]
    G = ctx.groups
    if cz == c:
        gam, bet = bn.weight.detach(), bn.bias.detach()
    else:
        gam = torch.zeros(cz, dtype=torch.float32, device=z.device)
        gam[:c] = bn.weight.detach()
        bet = torch.zeros(cz, dtype=torch.float32, device=z.device)
        bet[:c] = bn.bias.detach()
    # a unit without a residual input recomputes its ReLU mask from z (already read) instead of re-reading the 16-bit output y
    ymask = y if (ctx.has_res or not ctx.relu) else None
    sums = channel_sums(dy, ymask, z, ctx.mean, ctx.invstd, relu=ctx.relu, groups=G, gamma=gam, beta=bet, zcode=ctx.zcode)
    dz = Act.empty(n, t, h, w, cz, y.buf.dtype, z.device)
    dres = Act.empty(n, t, h, w, cz, y.buf.dtype, z.device) if ctx.has_res else None
    # (deterministic mode: the fused bias-gradient accumulation adds from many threads per address; ConvLayer.wgrad then forms it with the channel-sum kernel)
    dbias = ARENA.take((DB_SLOTS, cz), z.device) if (ctx.conv.bias is not None and not E.DETERMINISTIC) else None
    check(_lib.lib().tedspad_bn_bwd_apply(dy.ptr, ymask.ptr if ymask is not None else None, z.data_ptr(), ctx.zcode, ctx.mean.data_ptr(), ctx.invstd.data_ptr(), gam.data_ptr(),
                                          bet.data_ptr(), sums.data_ptr(), cz, dz.ptr, dres.ptr if dres is not None else None,
                                          dbias.data_ptr() if dbias is not None else None, DB_SLOTS, n * t * h * w // G, cz,
                                          dy.ld, y.ld, cz, dz.ld, dres.ld if dres is not None else 0, int(ctx.relu), G,
                                          _code(y.buf), _stream_ptr()), "tedspad_bn_bwd_apply")
    for sg in (sums.unbind(0) if G > 1 else (sums,)):                # d(beta), d(gamma): the groups' sums add up
        defer_grad(bn.bias, sg[0, :c])
        defer_grad(bn.weight, sg[1, :c])
    ctx.conv.wgrad(ctx.x, dz, db=dbias.sum(0) if dbias is not None else None)
    dx = None
    if need_dx:
        dx = ctx.conv.dgrad(dz, ctx.x.dims[1:] if x_dims is None else x_dims, residual=dx_residual, mask=dx_mask)
    return dx, dres


# ---- pooling / resize backward -----------------------------------------------------------------------

def maxpool_bwd(x: Act, idx: torch.Tensor, dy: Act, k, s, pads=(0, 0, 0), add: Optional[Act] = None, relu_mask=False) -> Act:
    """idx: the argmax tensor `engine.maxpool(..., return_idx=True)` recorded in the forward pass."""
    n, t, h, w = x.dims
    _, to, ho, wo = dy.dims
    dx = Act.empty(n, t, h, w, x.c, x.buf.dtype, x.buf.device)
    d = PoolDesc(n=n, t=t, h=h, w=w, c=x.c, ldx=x.ld, ldy=dy.ld, kt=k[0], kh=k[1], kw=k[2], st=s[0], sh=s[1], sw=s[2],
                 pt=pads[0], ph=pads[1], pw=pads[2], to=to, ho=ho, wo=wo, pad_zero=0, dtype=_code(x.buf))
    check(_lib.lib().tedspad_maxpool_bwd(C.byref(d), x.ptr, idx.data_ptr(), dy.ptr, dy.ld, add.ptr if add is not None else None,
                                         add.ld if add is not None else 0, dx.ptr, dx.ld, int(relu_mask), _stream_ptr()), "tedspad_maxpool_bwd")
    return dx


def global_avgpool_bwd(dfeat: torch.Tensor, like: Act, mask: Optional[Act] = None) -> Act:
    n, t, h, w = like.dims
    dx = Act.empty(n, t, h, w, like.c, like.buf.dtype, like.buf.device)
    dfeat = dfeat.contiguous().float()
    check(_lib.lib().tedspad_global_avgpool_bwd(dfeat.data_ptr(), mask.ptr if mask is not None else None, mask.ld if mask is not None else 0,
                                                dx.ptr, n, t * h * w, like.c, dx.ld, _code(like.buf), _stream_ptr()), "tedspad_global_avgpool_bwd")
    return dx


def upsample2x_bwd(dy_slice: Act, h: int, w: int, pad_top=0, pad_left=0) -> Act:
    n, _, ho, wo = dy_slice.dims
    dx = Act.empty(n, 1, h, w, dy_slice.c, dy_slice.buf.dtype, dy_slice.buf.device)
    check(_lib.lib().tedspad_upsample_bilinear2x_bwd(dy_slice.ptr, dx.ptr, n, h, w, dy_slice.c, dy_slice.ld, dx.ld, ho, wo, pad_top, pad_left,
                                                     _code(dx.buf), _stream_ptr()), "tedspad_upsample_bilinear2x_bwd")
    return dx


def nchw_grad_to_act(dy: torch.Tensor, y_sigmoid: Optional[torch.Tensor], dims, dtype=E.DEFAULT_DTYPE) -> Act:
    """fp32 (n,c,*spatial) gradient -> 16-bit (n,t,h,w,8); with y_sigmoid also the sigmoid backward."""
    n, c = dy.shape[:2]
    t, h, w = dims
    tdt, code = E.DTYPES[dtype]
    out = Act.empty(n, t, h, w, 8, tdt, dy.device)
    dy = dy.contiguous().float()
    ys = y_sigmoid.contiguous().float() if y_sigmoid is not None else None
    check(_lib.lib().tedspad_nchw_grad_to_channels_last(dy.data_ptr(), ys.data_ptr() if ys is not None else None, out.ptr, n, c, t * h * w,
                                                        code, _stream_ptr()), "tedspad_nchw_grad_to_channels_last")
    return out


def act_to_nchw_into(x: Act, c: int, dst: torch.Tensor):
    """x (n,t,h,w,ld) first c channels -> fp32 `dst` (n,c,t,h,w) view with arbitrary strides."""
    n, t, h, w = x.dims
    assert tuple(dst.shape) == (n, c, t, h, w) and dst.dtype == torch.float32
    sn, sc, st, sh, sw = dst.stride()
    check(_lib.lib().tedspad_channels_last_to_nchw_strided(x.ptr, dst.data_ptr(), n, c, t, h, w, x.ld, sn, sc, st, sh, sw, _code(x.buf),
                                                           _stream_ptr()), "tedspad_channels_last_to_nchw_strided")
    return dst
