"""-m gpu: the multi-job weight refresh (csrc/pack.hip: tedspad_pack_multi / tedspad_fold_multi / tedspad_wgrad_unpack_multi).
After an optimizer step the reference just reads its fp32 parameters again (train_anonymizer.py:123,193); here every 16-bit weight image
and folded BatchNorm vector of the updated network is rewritten in place by one launch per kind. The checks are bit-exact: a refreshed
image must equal the image a fresh pack of the same parameters gives, and the multi-job gradient unpack must equal the per-layer path."""
import pytest
import torch

from ted_spad_amd.synth import synth_tensor

pytestmark = pytest.mark.gpu
H = torch.float16

# name, (co, ci, kt, kh, kw), stride, pads, pair_w, bias, x dims (t, h, w)
LAYERS = [
    ("stem_pairs", (64, 3, 5, 7, 7), (2, 2, 2), (2, 3, 3), 3, False, (4, 16, 16)),
    ("3x3x3", (128, 64, 3, 3, 3), (1, 1, 1), (1, 1, 1), None, False, (2, 6, 6)),
    ("1x3x3_s2_ci40", (96, 40, 1, 3, 3), (1, 2, 2), (0, 1, 1), None, False, (2, 9, 9)),
    ("1x1_s2", (256, 128, 1, 1, 1), (1, 2, 2), (0, 0, 0), None, False, (2, 7, 7)),
    ("2d_bias", (24, 16, 3, 3), (1, 1, 1), (0, 1, 1), None, True, (1, 8, 8)),
]


def _layers():
    from ted_spad_amd import train_engine as TE
    from ted_spad_amd.params import BNParams
    out = []
    for i, (name, shape, stride, pads, pair_w, bias, xd) in enumerate(LAYERS):
        w = torch.nn.Parameter((synth_tensor(20 + i, name + "w", shape, -1, 1) * 0.1).cuda())
        b = torch.nn.Parameter(synth_tensor(20 + i, name + "b", (shape[0],), -0.2, 0.2).cuda()) if bias else None
        bn = BNParams(shape[0]).cuda()
        with torch.no_grad():
            bn.weight.copy_(synth_tensor(20 + i, "g", (shape[0],), 0.5, 1.5)); bn.bias.copy_(synth_tensor(20 + i, "be", (shape[0],), -0.3, 0.3))
            bn.running_mean.copy_(synth_tensor(20 + i, "m", (shape[0],), -0.2, 0.2)); bn.running_var.copy_(synth_tensor(20 + i, "v", (shape[0],), 0.5, 2.0))
        out.append((TE.ConvLayer(w, b, stride, pads, pair_w=pair_w), bn, xd))
    return out


def _acts(L, xd, n=2):
    """(x, dy) Acts of the right shapes for layer L."""
    from ted_spad_amd import engine as E
    co, ci = L.weight.shape[:2]
    k = tuple(L._w5().shape[2:])
    od = tuple(E.conv_out(xd[i], k[i], L.stride[i], L.pads[i], L.pads_back[i]) for i in range(3))
    if L.pair_w is not None:
        x = E.clip_to_act(synth_tensor(3, "x", (n, ci) + xd, -1, 1).cuda(), cpad=4, dtype="f16")
    else:
        cin = (ci + 7) // 8 * 8
        xb = torch.zeros((n,) + xd + (cin,), dtype=H, device="cuda")
        xb[..., :ci] = synth_tensor(3, "x", (n,) + xd + (ci,), -1, 1).to(H).cuda()
        x = E.Act(xb, cin)
    c8 = (co + 7) // 8 * 8
    db = torch.zeros((n,) + od + (c8,), dtype=H, device="cuda")
    db[..., :co] = synth_tensor(3, "dy", (n,) + od + (co,), -1, 1).to(H).cuda()
    return x, E.Act(db, c8)


def test_refreshed_images_equal_a_fresh_pack():
    """Every image a ConvLayer holds (train-flavour and folded forward images, data-gradient images of both flavours, incl. the stem's
    pixel-pair form, strided parity classes and a channel count that is no multiple of 8) after parameters AND BatchNorm tensors changed
    in place: WeightRefresh.run() (2 launches) == images built from scratch, bit for bit; the lazy path then has nothing to rebuild."""
    from ted_spad_amd import engine as E, train_engine as TE
    items = _layers()
    folds = {}
    layers = [L for L, _, _ in items]
    for L, bn, xd in items:
        x, dy = _acts(L, xd)
        s, b = TE.cached_fold(folds, bn, L.bias)
        L.forward(x)
        L.forward(x, scale=s, shift=b)
        L.dgrad(dy, x.dims[1:])
        L.dgrad(dy, x.dims[1:], scale=s)
    R = TE.WeightRefresh(lambda: layers, folds)
    assert R.run() is False                                   # nothing changed yet
    with torch.no_grad():
        for i, (L, bn, _) in enumerate(items):
            L.weight.mul_(1.25).add_(0.01 * (i + 1))
            if L.bias is not None:
                L.bias.add_(0.05)
            bn.weight.mul_(0.9); bn.bias.add_(0.1); bn.running_mean.add_(0.03); bn.running_var.mul_(1.1)
    gen = TE.IMAGES_GEN
    assert R.stale(layers) and R.run() is True
    torch.cuda.synchronize()
    nimg = 0
    for L, bn, xd in items:
        s_new, b_new = E.fold_bn(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, conv_bias=L.bias)
        s, b = TE.cached_fold(folds, bn, L.bias)
        assert torch.equal(s, s_new) and torch.equal(b, b_new)
        w5 = L._w5()
        for (s_none, b_none), (_, pc, sc, sh) in L._fwd.items():
            fresh = E.PackedConv(w5, sc, (L.bias.detach() if L.bias is not None else None) if sh is None else sh, stride=L.stride, dtype=L.dtype, pair_w=L.pair_w)
            assert torch.equal(pc.w, fresh.w), "forward image"
            assert torch.equal(pc.scale, fresh.scale) and torch.equal(pc.shift, fresh.shift), "scale / shift vectors"
            if not s_none:
                assert torch.equal(pc.scale[:s_new.numel()], s_new) and float(pc.scale[s_new.numel():].abs().sum()) == 0
            nimg += 1
        for (x_dims, dy_dims, s_none), (_, plan, sc) in L._dgrad.items():
            pk, _ = L._pads_k(L.geom_conv())
            fresh = TE.DgradPlan(w5.float(), sc, L.geom_conv().stride, pk, x_dims, dy_dims, L.dtype, pair_w=L.pair_w)
            assert len(fresh.subs) == len(plan.subs)
            for (_, a, _, _), (_, f, _, _) in zip(plan.subs, fresh.subs):
                assert torch.equal(a.w, f.w), "data-gradient image"
                nimg += 1
    assert nimg >= 4 * len(items)
    x, dy = _acts(items[1][0], items[1][2])
    items[1][0].forward(x); items[1][0].dgrad(dy, x.dims[1:])
    assert TE.IMAGES_GEN == gen and R.run() is False           # the lazy path found every image fresh


def test_refresh_follows_parameters_whose_storage_moved():
    """A parameter or BatchNorm tensor that MOVED (`p.data = ...`, a `.to()` / dtype round trip, an EMA swap) since the job tables were built: the refresh must
    read the new storage -- the cached tables hold the old addresses (possibly freed memory) -- and the images must equal a fresh pack of the new values
    (round-2 advisor finding: the tables were reused and the stale images then stamped as fresh)."""
    from ted_spad_amd import engine as E, train_engine as TE
    items = _layers()[1:4]
    folds = {}
    layers = [L for L, _, _ in items]
    for L, bn, xd in items:
        x, dy = _acts(L, xd)
        s, b = TE.cached_fold(folds, bn, L.bias)
        L.forward(x); L.forward(x, scale=s, shift=b); L.dgrad(dy, x.dims[1:]); L.dgrad(dy, x.dims[1:], scale=s)
    R = TE.WeightRefresh(lambda: layers, folds)
    with torch.no_grad():
        for L, _, _ in items:
            L.weight.mul_(1.1)
    assert R.run() is True                                    # tables built on the original storage
    old = [(L.weight.data, bn.running_var.data) for L, bn, _ in items]
    with torch.no_grad():
        for i, (L, bn, _) in enumerate(items):
            L.weight.data = (L.weight.data * 0.5 + 0.02 * (i + 1)).clone()       # new storage, new values
            bn.running_var.data = (bn.running_var.data * 1.3).clone()
    for w_old, v_old in old:                                   # what a stale table would read
        w_old.fill_(float("nan")); v_old.fill_(float("nan"))
    assert R.run() is True
    torch.cuda.synchronize()
    for L, bn, xd in items:
        s_new, b_new = E.fold_bn(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, conv_bias=L.bias)
        s, b = TE.cached_fold(folds, bn, L.bias)
        assert torch.equal(s, s_new) and torch.equal(b, b_new)
        w5 = L._w5()
        for (s_none, b_none), (_, pc, sc, sh) in L._fwd.items():
            fresh = E.PackedConv(w5, sc, (L.bias.detach() if L.bias is not None else None) if sh is None else sh, stride=L.stride, dtype=L.dtype, pair_w=L.pair_w)
            assert torch.equal(pc.w, fresh.w) and not bool(torch.isnan(pc.w.float()).any()), "forward image read the old storage"
        for (x_dims, dy_dims, s_none), (_, plan, sc) in L._dgrad.items():
            pk, _ = L._pads_k(L.geom_conv())
            fresh = TE.DgradPlan(w5.float(), sc, L.geom_conv().stride, pk, x_dims, dy_dims, L.dtype, pair_w=L.pair_w)
            for (_, a, _, _), (_, f, _, _) in zip(plan.subs, fresh.subs):
                assert torch.equal(a.w, f.w), "data-gradient image read the old storage"
    assert R.run() is False


def test_running_statistics_written_by_the_kernel_refold_the_frozen_flavour():
    """tedspad_bn_train_apply updates running_mean / running_var through raw pointers (no version bump): the fold signature follows
    num_batches_tracked instead, so an eval-flavour forward after a train-mode forward WITHOUT an optimizer step sees the new statistics."""
    from ted_spad_amd import train_engine as TE
    L, bn, xd = _layers()[1]
    folds = {}
    x, _ = _acts(L, xd)
    s0 = TE.cached_fold(folds, bn)[0].clone()
    TE.conv_bn_act_train(L, bn, x)
    TE.flush_deferred()
    torch.cuda.synchronize()
    s1 = TE.cached_fold(folds, bn)[0]
    assert not torch.equal(s0, s1), "the fold still holds the statistics from before the train-mode forward"


def test_stale_fold_alone_marks_the_images_stale():
    """Only a BatchNorm changed (weights untouched): cached_fold re-folds into the same vectors and the data-gradient image that has the old
    scale folded in is rebuilt."""
    from ted_spad_amd import train_engine as TE
    L, bn, xd = _layers()[1]
    folds = {}
    x, dy = _acts(L, xd)
    s, _ = TE.cached_fold(folds, bn)
    d0 = L.dgrad(dy, x.dims[1:], scale=s).buf.float().clone()
    with torch.no_grad():
        bn.weight.mul_(2.0)
    s2, _ = TE.cached_fold(folds, bn)
    assert s2.data_ptr() == s.data_ptr()
    d1 = L.dgrad(dy, x.dims[1:], scale=s2).buf.float()
    assert float((d1 - 2 * d0).abs().max()) <= 2e-3 * float(d0.abs().max())


@pytest.mark.parametrize("row_scale", [False, True])
def test_multi_job_gradient_unpack_equals_the_per_layer_path(row_scale):
    """flush_conv_grads (one tedspad_wgrad_unpack_multi launch + the stem on its own path) against the per-layer flush_grad on copies of the
    same packed accumulators: first call creates .grad, the second accumulates into it (cached job table); with the frozen-BatchNorm row scale."""
    from ted_spad_amd import train_engine as TE
    items = _layers()
    layers = [L for L, _, _ in items]
    dev = layers[0].weight.device
    ref = [None] * len(layers)
    refb = [None] * len(layers)
    for rnd in range(2):
        TE.ARENA.reset(dev)
        scales = []
        for i, (L, bn, xd) in enumerate(items):
            x, dy = _acts(L, xd)
            L.wgrad(x, dy)
            L.wgrad(x, dy)                                     # two contributions per step, as the three clips give
            rs = synth_tensor(5, "rs%d" % i, (L.weight.shape[0],), 0.5, 1.5).cuda() if row_scale else None
            scales.append(rs)
        torch.cuda.synchronize()
        # the per-layer path on clones of the accumulators, into separate gradient tensors
        for i, L in enumerate(layers):
            keep_w, keep_b, dwp, db = L.weight.grad, (L.bias.grad if L.bias is not None else None), L._dwp, L._db
            L.weight.grad = ref[i]
            if L.bias is not None:
                L.bias.grad = refb[i]
            L._dwp = dwp.clone()
            L._grad_row_scale = scales[i]
            L.flush_grad()
            ref[i] = L.weight.grad
            if L.bias is not None:
                refb[i] = L.bias.grad
                L.bias.grad = keep_b
            L.weight.grad, L._dwp, L._db, L._dwp_gen = keep_w, dwp, db, TE.ARENA.gen
            L._grad_row_scale = scales[i]
        TE.flush_conv_grads(layers)
        torch.cuda.synchronize()
        for i, L in enumerate(layers):
            assert L._dwp is None
            assert torch.equal(L.weight.grad, ref[i]), (rnd, LAYERS[i][0])
            if L.bias is not None:
                assert torch.equal(L.bias.grad, refb[i])
    assert len(TE._UNPACK_TABLES) >= 1


def _flat(P):
    out = []
    for v in P.values():
        out += list(v) if isinstance(v, (list, tuple)) else [v]
    return out


def test_eval_networks_refresh_in_place_after_an_in_place_update(monkeypatch):
    """UNet / UnetPlusPlus in eval mode: after an in-place parameter + running-statistics update `packed()` keeps its PackedConv objects
    (tuned tiles, gather tables) and the output equals a freshly built copy of the network with the same state."""
    from ted_spad_amd.model_loaders import load_fa_model
    from ted_spad_amd.synth import synth_state_dict
    from ted_spad_amd import engine as E
    import contextlib, io
    monkeypatch.setattr(E, "AUTOTUNE", False)                  # one tile per conv: the two networks run the same launches
    x = synth_tensor(6, "img", (2, 3, 32, 32), 0, 1).cuda()
    for arch in ("unet", "unet++"):
        with contextlib.redirect_stdout(io.StringIO()):
            m, m2 = load_fa_model(arch=arch), load_fa_model(arch=arch)
        m.load_state_dict(synth_state_dict(m.state_dict(), 1))
        m = m.cuda().eval()
        with torch.no_grad():
            m(x)
            ids = [id(v) for v in _flat(m.packed())]
            imgs = [v.w.clone() for v in _flat(m.packed())]
            for p in m.parameters():
                p.mul_(1.1)
            for name, b in m.named_buffers():
                if name.endswith("running_var"):
                    b.mul_(1.3)
                elif name.endswith("running_mean"):
                    b.add_(0.02)
            y = m(x)
            assert [id(v) for v in _flat(m.packed())] == ids and len(ids) >= 19, "objects were rebuilt"
            assert all(not torch.equal(a, v.w) for a, v in zip(imgs, _flat(m.packed()))), "an image was not rewritten"
            m2.load_state_dict(m.state_dict())
            y2 = m2.cuda().eval()(x)
        assert torch.equal(y, y2), arch


def test_fused_optimizer_steps_need_mark_updated():
    """torch's fused optimizers write the parameters without bumping their version counters (pinned here: if a later torch does bump them this test says so and the
    helper becomes a no-op): the refresh cannot see such a step until `train_engine.mark_updated` says the parameters changed; then the rewritten image equals a
    fresh pack."""
    from ted_spad_amd import engine as E, train_engine as TE
    L, bn, xd = _layers()[1]
    x, _ = _acts(L, xd)
    pc = L.fwd_conv()
    refresh = TE.WeightRefresh(lambda: [L])
    refresh.run()
    assert not refresh.stale([L])
    opt = torch.optim.Adam([L.weight], lr=0.05, fused=True)
    L.weight.grad = torch.ones_like(L.weight)
    v0, w0 = L.weight._version, L.weight.detach().clone()
    opt.step()
    assert not torch.equal(L.weight.detach(), w0)
    if L.weight._version == v0:
        assert not refresh.stale([L])                      # the step is invisible to the version counter ...
        TE.mark_updated([L.weight])
    assert refresh.stale([L])                              # ... and visible once marked
    refresh.run()
    torch.cuda.synchronize()
    fresh = E.PackedConv(L._w5(), None, None, stride=L.stride, dtype=L.dtype, pair_w=L.pair_w)
    assert torch.equal(L.fwd_conv().w, fresh.w) and L.fwd_conv() is pc

