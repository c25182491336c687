"""-m gpu: the training-path kernels (data gradient, weight gradient, train-mode BatchNorm, pooling /
upsample backward) against torch CPU autograd of the reference's own ops (fp32/fp64) on seeded inputs
pre-rounded to f16, so only summation order and output rounding differ."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2
from ted_spad_amd.synth import synth_tensor

pytestmark = pytest.mark.gpu
H = torch.float16


def cl(t):  # (n,c,t,h,w) fp32 cpu -> channels-last f16 Act on the GPU
    from ted_spad_amd import engine as E
    return E.Act(t.permute(0, 2, 3, 4, 1).contiguous().to(H).cuda(), t.shape[1])


def nc(a, c=None):  # Act -> (n,c,t,h,w) fp32 cpu
    x = a.buf.float().cpu()
    x = x[..., a.coff:a.coff + (a.c if c is None else c)]
    return x.permute(0, 4, 1, 2, 3)


CONVS = [
    # name, cin, cout, k, stride, pads, (t,h,w)
    ("1x1", 64, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 9, 9)),
    ("3x1x1", 128, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), (4, 7, 7)),
    ("1x3x3", 64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (2, 12, 11)),
    ("1x3x3_s2_odd", 64, 128, (1, 3, 3), (1, 2, 2), (0, 1, 1), (2, 15, 15)),
    ("1x1_s2_odd", 64, 128, (1, 1, 1), (1, 2, 2), (0, 0, 0), (2, 15, 15)),
    ("1x1_s2_even", 64, 128, (1, 1, 1), (1, 2, 2), (0, 0, 0), (2, 14, 14)),
    ("unet3x3", 128, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 28, 28)),
    # the three-tile weight-gradient kernel (conv_wgrad3_kernel, rows of >= 66 pixels) and the gather kernel on 3 x 3 layers it does not take:
    # a partial 64-channel co tile, two ci chunks, frames as independent images
    ("3x3_w72", 64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 9, 72)),
    ("3x3_w80_frames_co96_c128", 128, 96, (1, 3, 3), (1, 1, 1), (0, 1, 1), (3, 5, 80)),       # six 5-row images: rows above / below missing often
    ("3x3_co96_w7", 64, 96, (1, 3, 3), (1, 1, 1), (0, 1, 1), (2, 7, 7)),
    ("3x3_c128_co192", 128, 192, (1, 3, 3), (1, 1, 1), (0, 1, 1), (2, 10, 20)),
]


@pytest.mark.parametrize("case", CONVS, ids=[c[0] for c in CONVS])
def test_dgrad_and_wgrad_vs_autograd(case):
    from ted_spad_amd import train_engine as TE
    name, cin, cout, k, stride, pads, thw = case
    x = synth_tensor(1, name + "x", (2, cin) + thw, -1, 1).to(H).float().requires_grad_()
    w = (synth_tensor(1, name + "w", (cout, cin) + k, -1, 1) * (1.0 / (cin * k[0] * k[1] * k[2])) ** 0.5).to(H).float().requires_grad_()
    y = F.conv3d(x, w, stride=stride, padding=pads)
    dy = synth_tensor(1, name + "dy", tuple(y.shape), -1, 1).to(H).float()
    res = synth_tensor(1, name + "res", tuple(x.shape), -1, 1).to(H).float()
    y.backward(dy)
    wp = torch.nn.Parameter(w.detach().clone().cuda())
    layer = TE.ConvLayer(wp, None, stride, pads)
    # data gradient, with the fused (+residual) * [mask > 0] epilogue
    if stride == (1, 1, 1):
        dx = layer.dgrad(cl(dy), thw, residual=cl(res), mask=cl(x.detach()))
        assert rel_l2(nc(dx), (x.grad + res) * (x.detach() > 0)) < 1e-3
    else:   # strided: some input positions are never read by the forward conv -> masked only
        dx = layer.dgrad(cl(dy), thw, mask=cl(x.detach()))
        assert rel_l2(nc(dx), x.grad * (x.detach() > 0)) < 1e-3
    dx2 = layer.dgrad(cl(dy), thw)
    assert rel_l2(nc(dx2), x.grad) < 1e-3
    # weight gradient
    layer.wgrad(cl(x.detach()), cl(dy))
    layer.flush_grad()
    assert rel_l2(wp.grad.cpu(), w.grad) < 1e-3


def test_stem_pair_form_backward():
    """5x7x7 stride-2 stem (Cin=3): forward in pixel-pair form, dgrad gives d(clip) as (n,t,h,w,4), wgrad in (co,3,5,7,7)."""
    from ted_spad_amd import engine as E, train_engine as TE
    x = synth_tensor(2, "sx", (2, 3, 8, 32, 32)).to(H).float().requires_grad_()
    w = (synth_tensor(2, "sw", (64, 3, 5, 7, 7), -1, 1) * 0.05).to(H).float().requires_grad_()
    y = F.conv3d(x, w, stride=2, padding=(2, 3, 3))
    dy = synth_tensor(2, "sdy", tuple(y.shape), -1, 1).to(H).float()
    y.backward(dy)
    wp = torch.nn.Parameter(w.detach().clone().cuda())
    layer = TE.ConvLayer(wp, None, (2, 2, 2), (2, 3, 3), pair_w=3)
    xa = E.clip_to_act(x.detach().cuda(), cpad=4)
    out = layer.forward(xa)
    assert rel_l2(nc(out), y.detach()) < 1e-3
    dxp = layer.dgrad(cl(dy), xa.dims[1:])                       # (2,8,32,16,8) == (2,8,32,32,4)
    dx = dxp.buf.float().cpu().reshape(2, 8, 32, 32, 4)[..., :3].permute(0, 4, 1, 2, 3)
    assert rel_l2(dx, x.grad) < 1e-3
    layer.wgrad(xa, cl(dy))
    layer.flush_grad()
    assert rel_l2(wp.grad.cpu(), w.grad) < 1e-3


@pytest.mark.parametrize("z16", [True, False], ids=["z16", "z32"])
@pytest.mark.parametrize("with_res", [False, True])
def test_conv_bn_relu_train_fwd_bwd(with_res, z16, monkeypatch):
    from ted_spad_amd import engine as E, train_engine as TE
    monkeypatch.setattr(TE, "TRAIN_Z16", z16)       # the conv output in front of the BatchNorm in the 16-bit activation dtype (default) or fp32
    from ted_spad_amd.params import BNParams
    cin, cout, thw = 64, 128, (2, 9, 10)
    x = synth_tensor(3, "x", (3, cin) + thw, -1, 1).to(H).float().requires_grad_()
    w = (synth_tensor(3, "w", (cout, cin, 1, 3, 3), -1, 1) * 0.06).to(H).float().requires_grad_()
    b = synth_tensor(3, "b", (cout,), -0.5, 0.5).requires_grad_()
    g = synth_tensor(3, "g", (cout,), 0.5, 1.5).requires_grad_()
    be = synth_tensor(3, "be", (cout,), -0.3, 0.3).requires_grad_()
    rm, rv = synth_tensor(3, "rm", (cout,), -0.1, 0.1), synth_tensor(3, "rv", (cout,), 0.5, 1.5)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    z = F.conv3d(x, w, b, padding=(0, 1, 1))
    u = F.batch_norm(z, rm_ref, rv_ref, g, be, training=True, momentum=0.1, eps=1e-5)
    res = synth_tensor(3, "res", tuple(u.shape), -1, 1).to(H).float().requires_grad_() if with_res else None
    y = F.relu(u + res if with_res else u)
    dy = synth_tensor(3, "dy", tuple(y.shape), -1, 1).to(H).float()
    # The ReLU gradient is discontinuous at 0: with the pre-activation stored in 16 bits, elements within one
    # rounding step of 0 take the other branch than in fp32 (each flip is an O(1) difference; ~2e-4 of the
    # elements -> ~1.5e-2 rel-L2). No gradient is sent through that band so the test checks the arithmetic.
    pre = (u + res if with_res else u).detach()
    dy = dy * (pre.abs() > 0.02)
    y.backward(dy)
    bn = BNParams(cout).cuda()
    with torch.no_grad():
        bn.weight.copy_(g); bn.bias.copy_(be); bn.running_mean.copy_(rm); bn.running_var.copy_(rv)
    wp, bp = torch.nn.Parameter(w.detach().clone().cuda()), torch.nn.Parameter(b.detach().clone().cuda())
    layer = TE.ConvLayer(wp, bp, (1, 1, 1), (0, 1, 1))
    ya, ctx = TE.conv_bn_act_train(layer, bn, cl(x.detach()), relu=True, residual=cl(res.detach()) if with_res else None)
    assert rel_l2(nc(ya), y.detach()) < 2e-3
    assert rel_l2(bn.running_mean.cpu(), rm_ref) < 1e-3 and rel_l2(bn.running_var.cpu(), rv_ref) < 1e-3
    TE.flush_deferred()      # counters / BN gradients are applied in multi-tensor batches
    assert int(bn.num_batches_tracked) == 1
    seen, wgrad = {}, layer.wgrad

    def spy(xa, dza, db=None):
        seen["dz"], seen["db"] = dza, db
        return wgrad(xa, dza, db=db)
    layer.wgrad = spy
    dx, dres = TE.conv_bn_act_train_bwd(ctx, cl(dy))
    layer.flush_grad()
    assert rel_l2(nc(dx), x.grad) < 5e-3
    assert rel_l2(wp.grad.cpu(), w.grad) < 5e-3
    assert rel_l2(bn.weight.grad.cpu(), g.grad) < 5e-3 and rel_l2(bn.bias.grad.cpu(), be.grad) < 5e-3
    assert float(bp.grad.abs().max()) < 2e-2 * float(dy.abs().sum()) ** 0.5   # analytically 0 (BN removes the mean)
    # the conv-bias gradient is gathered by the BatchNorm backward while it writes dz: == the per-channel sum of dz up to dz's 16-bit rounding
    # (in deterministic mode ConvLayer.wgrad forms it with the channel-sum kernel instead and receives no `db`: the check below holds either way)
    dz = nc(seen["dz"])
    assert (seen["db"] is None) == E.DETERMINISTIC
    npx = dz.numel() // cout
    tol = 4 * npx ** 0.5 * float(dz.pow(2).mean().sqrt()) * 2.0 ** -11 + 1e-6
    assert float((bp.grad.cpu() - dz.sum(dim=(0, 2, 3, 4))).abs().max()) < tol
    if with_res:
        assert rel_l2(nc(dres), res.grad) < 1e-3


@pytest.mark.parametrize("k,cfg", [((1, 3, 3), None), ((1, 1, 1), None), ((1, 3, 3), 32), ((1, 3, 3), 33), ((1, 3, 3), 3), ((1, 1, 1), 1)])
def test_conv_bn_relu_train_grouped_statistics(k, cfg, monkeypatch):
    """groups = 3: ONE launch sequence over a batch of three blocks of samples, each normalised with its own batch statistics and the
    running statistics updated block after block -- against torch calling conv + BatchNorm3d(train) three times (the three clips of a
    training iteration, train_anonymizer.py:169-175). 576 rows per group: the 256-row tiles straddle the group boundaries."""
    from ted_spad_amd import engine as E, train_engine as TE
    from ted_spad_amd.params import BNParams
    monkeypatch.setattr(E, "FORCE_TILE_CFG", cfg)        # the generic tiles (64 .. 256 rows) and the chunk-major patch / flat tiles: both epilogues
    cin, cout, thw, G = 64, 128, (2, 12, 12), 3
    pad = (0, k[1] // 2, k[2] // 2)
    x = synth_tensor(4, "gx", (2 * G, cin) + thw, -1, 1).to(H).float().requires_grad_()
    w = (synth_tensor(4, "gw", (cout, cin) + k, -1, 1) * (0.06 if k[1] == 3 else 0.15)).to(H).float().requires_grad_()
    g = synth_tensor(4, "gg", (cout,), 0.5, 1.5).requires_grad_()
    be = synth_tensor(4, "gbe", (cout,), -0.3, 0.3).requires_grad_()
    rm, rv = synth_tensor(4, "grm", (cout,), -0.1, 0.1), synth_tensor(4, "grv", (cout,), 0.5, 1.5)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    gain = torch.tensor([1.0, 0.5, 2.0]).repeat_interleave(2).view(-1, 1, 1, 1, 1)          # the groups have different statistics
    z = F.conv3d(x * gain, w, None, padding=pad)
    u = torch.cat([F.batch_norm(z[2 * i:2 * i + 2], rm_ref, rv_ref, g, be, training=True, momentum=0.1, eps=1e-5) for i in range(G)])
    res = synth_tensor(4, "gres", tuple(u.shape), -1, 1).to(H).float().requires_grad_()
    y = F.relu(u + res)
    dy = synth_tensor(4, "gdy", tuple(y.shape), -1, 1).to(H).float() * ((u + res).detach().abs() > 0.02)
    y.backward(dy)
    bn = BNParams(cout).cuda()
    with torch.no_grad():
        bn.weight.copy_(g); bn.bias.copy_(be); bn.running_mean.copy_(rm); bn.running_var.copy_(rv)
    wp = torch.nn.Parameter(w.detach().clone().cuda())
    layer = TE.ConvLayer(wp, None, (1, 1, 1), pad)
    xin = cl((x.detach() * gain).to(H).float())
    ya, ctx = TE.conv_bn_act_train(layer, bn, xin, relu=True, residual=cl(res.detach()), groups=G)
    assert rel_l2(nc(ya), y.detach()) < 2e-3
    for i in range(G):                                   # every group on its own: a wrong group boundary shows here first
        assert rel_l2(nc(ya)[2 * i:2 * i + 2], y.detach()[2 * i:2 * i + 2]) < 2e-3, i
    assert rel_l2(bn.running_mean.cpu(), rm_ref) < 1e-3 and rel_l2(bn.running_var.cpu(), rv_ref) < 1e-3
    TE.flush_deferred()
    assert int(bn.num_batches_tracked) == G
    dx, dres = TE.conv_bn_act_train_bwd(ctx, cl(dy))
    layer.flush_grad()
    TE.flush_deferred()
    xg = x.grad / gain                                    # d/d(x * gain)
    assert rel_l2(nc(dx), xg) < 5e-3
    assert rel_l2(wp.grad.cpu(), w.grad) < 5e-3
    assert rel_l2(bn.weight.grad.cpu(), g.grad) < 5e-3 and rel_l2(bn.bias.grad.cpu(), be.grad) < 5e-3
    assert rel_l2(nc(dres), res.grad) < 1e-3


def test_pool_and_upsample_backward():
    from ted_spad_amd import engine as E, train_engine as TE
    # max-pool (2,3,3)/(2,2,2) with overlapping windows + accumulated extra gradient + ReLU mask
    x = synth_tensor(4, "px", (2, 64, 4, 13, 13), -1, 1).to(H).float().requires_grad_()
    y = F.max_pool3d(x, (2, 3, 3), (2, 2, 2))
    dy = synth_tensor(4, "pdy", tuple(y.shape), -1, 1).to(H).float()
    add = synth_tensor(4, "padd", tuple(x.shape), -1, 1).to(H).float()
    y.backward(dy)
    xa = cl(x.detach())
    ya, idx = E.maxpool(xa, (2, 3, 3), (2, 2, 2), return_idx=True)
    dx = TE.maxpool_bwd(xa, idx, cl(dy), (2, 3, 3), (2, 2, 2), add=cl(add), relu_mask=True)
    assert rel_l2(nc(dx), (x.grad + add) * (x.detach() > 0)) < 1e-3
    # global average pool
    df = synth_tensor(4, "df", (3, 128), -1, 1)
    like = cl(torch.zeros(3, 128, 2, 3, 3))
    m = synth_tensor(4, "am", (3, 128, 2, 3, 3), -1, 1)
    got = nc(TE.global_avgpool_bwd(df.cuda(), like, mask=cl(m)))
    assert rel_l2(got, (df.view(3, 128, 1, 1, 1) / 18).expand(3, 128, 2, 3, 3) * (m.to(H).float() > 0)) < 1e-3
    # bilinear x2 align_corners=True into a padded, wider buffer (odd skip size)
    for h, w, ho, wo in ((7, 7, 14, 14), (6, 5, 13, 11), (1, 3, 2, 6), (28, 28, 56, 56), (56, 56, 112, 112)):
        u = synth_tensor(4, "ux%d" % h, (2, 16, h, w), -1, 1).to(H).float().requires_grad_()
        up = F.interpolate(u, scale_factor=2, mode="bilinear", align_corners=True)
        dyy, dxx = ho - 2 * h, wo - 2 * w
        upp = F.pad(up, [dxx // 2, dxx - dxx // 2, dyy // 2, dyy - dyy // 2])
        g = synth_tensor(4, "ug%d" % h, tuple(upp.shape), -1, 1).to(H).float()
        upp.backward(g)
        ga = E.Act(g.permute(0, 2, 3, 1).unsqueeze(1).contiguous().to(H).cuda(), 16)
        du = TE.upsample2x_bwd(ga, h, w, dyy // 2, dxx // 2)
        assert rel_l2(du.buf.float().cpu()[:, 0].permute(0, 3, 1, 2), u.grad) < 2e-3
    # sigmoid backward + NCHW -> channels-last
    yv = torch.sigmoid(synth_tensor(4, "sy", (2, 3, 5, 6), -2, 2))
    gy = synth_tensor(4, "sg", (2, 3, 5, 6), -1, 1)
    a = TE.nchw_grad_to_act(gy.cuda(), yv.cuda(), (1, 5, 6))
    got = a.buf.float().cpu()[:, 0, :, :, :3].permute(0, 3, 1, 2)
    assert rel_l2(got, gy * yv * (1 - yv)) < 1e-3 and float(a.buf[..., 3:].abs().max()) == 0.0


def test_batch_chunking_matches_single_launch(monkeypatch):
    """Tensors beyond the kernels' 32-bit offsets (cfg5: 384 frames of 224x224) are processed in chunks of whole
    samples (engine.batch_chunk). With the limits lowered so that a 6-sample batch splits into 3 + 3 (forward / data
    gradient) and 2 + 2 + 2 (weight gradient), the results must equal the single-launch ones: bit-exact outputs,
    BatchNorm statistics and weight gradients up to float-atomic order (1e-5)."""
    from ted_spad_amd import engine as E, train_engine as TE
    n, cin, cout = 6, 16, 24
    x = cl(synth_tensor(9, "cx", (n, cin, 1, 12, 10), -1, 1))
    dy = cl(synth_tensor(9, "cdy", (n, cout, 1, 12, 10), -1, 1))
    res = cl(synth_tensor(9, "cr", (n, cout, 1, 12, 10), -1, 1))
    wp = torch.nn.Parameter(synth_tensor(9, "cw", (cout, cin, 3, 3), -0.2, 0.2).cuda())

    def run():
        layer = TE.ConvLayer(wp, None, (1, 1, 1), (0, 1, 1))
        stats = torch.zeros((2, layer.fwd_conv().cpad), device="cuda")
        y = layer.forward(x, relu=True, residual=res, stats=stats)
        dx = layer.dgrad(dy, x.dims[1:], mask=x)
        wp.grad = None
        layer.wgrad(x, dy)
        layer.flush_grad()
        return y.buf.clone(), stats.clone(), dx.buf.clone(), wp.grad.clone()

    one = run()
    per_sample = 12 * 10 * 24
    monkeypatch.setattr(E, "MAX_ELEMS", 3 * per_sample + 1)
    monkeypatch.setattr(E, "MAX_WGRAD_PIXELS", 2 * 12 * 10 + 1)
    assert E.batch_chunk(n, [per_sample], E.MAX_ELEMS) == 3
    many = run()
    assert torch.equal(one[0], many[0]) and torch.equal(one[2], many[2])
    assert torch.allclose(one[1], many[1], rtol=1e-5, atol=1e-5) and torch.allclose(one[3], many[3], rtol=1e-5, atol=1e-5)
    with pytest.raises(Exception):
        E.batch_chunk(2, [1 << 31], E.MAX_ELEMS)          # a single sample that does not fit fails loudly



@pytest.mark.parametrize("k,pads,pair", [((1, 3, 3), (0, 1, 1), False), ((5, 7, 7), (2, 3, 3), True)])
def test_stem_halo_tiles_gather_batch_statistics(k, pads, pair):
    """The halo-direct stem tiles (tile_cfg 9 / 20 / 21 / 29-31: cin = 8 records) with the batch-statistics epilogue a train-mode BatchNorm
    behind them needs -- the UNet's first conv (3 -> 64, 3 x 3) and I3Res50's stem in pixel-pair form -- against a generic tile on the same
    launch: outputs, plain sums and sums per statistics group (three groups of two samples)."""
    from ted_spad_amd import _lib, engine as E
    n, t, h, w = 6, (4 if pair else 1), 20, 24
    wgt = (synth_tensor(51, "shw", (64, 3) + k, -1, 1) * 0.1).to(H).float()
    if pair:
        pc = E.PackedConv(wgt, None, None, stride=(2, 2, 2), dtype="f16", device="cuda", pair_w=3)
        x = E.clip_to_act(synth_tensor(51, "shx", (n, 3, t, h, w), -1, 1).cuda(), cpad=4, dtype="f16")
        kw = dict(pads=(2, 3, pc.pair_pw), pads_back=(2, 3, 1))
    else:
        pc = E.PackedConv(wgt, None, None, dtype="f16", device="cuda")
        xb = torch.zeros((n, t, h, w, 8), dtype=H, device="cuda")
        xb[..., :3] = synth_tensor(51, "shx", (n, t, h, w, 3), -1, 1).to(H).cuda()
        x = E.Act(xb, 8)
        kw = dict(pads=pads)
    got = {}
    try:
        for cfg in (5, 9, 20, 21, 29, 30, 31):
            E.FORCE_TILE_CFG = cfg
            try:
                st, sg = torch.zeros((2, pc.cpad), device="cuda"), torch.zeros((3, 2, pc.cpad), device="cuda")
                y = pc(x, stats=st, **kw)
                pc(x, stats=sg, **kw)                          # (G, 2, ld): three groups of two samples
                got[cfg] = (y.buf.float().cpu(), st.cpu(), sg.cpu())
            except _lib.TedSpadHipError:
                continue
    finally:
        E.FORCE_TILE_CFG = None
    assert 5 in got and len(got) >= 3, sorted(got)
    ref = got[5]
    for cfg, (y, st, sg) in got.items():
        assert bool(((y - ref[0]).abs() <= 2.0 ** -10 * ref[0].abs() + 1e-4).all()), cfg
        assert rel_l2(st, ref[1]) < 1e-5, cfg
        assert rel_l2(sg, ref[2]) < 1e-5, cfg
        assert rel_l2(sg.sum(0), st) < 1e-5, cfg


def test_three_tile_weight_gradient_on_seeded_random_geometries():
    """conv_wgrad3_kernel (every 1 x 3 x 3 stride-1 'same' conv with whole 64-channel chunks) against torch autograd on seeded random shapes: rows
    narrower and wider than a 64-pixel step, single rows and columns, several frames per sample (independent images), ragged cout, 1-3 ci chunks,
    pixel counts that end inside a step and inside a split."""
    from ted_spad_amd import train_engine as TE
    rng = np.random.RandomState(4321)
    for case in range(8):
        n, t = int(rng.randint(1, 4)), int(rng.randint(1, 4))
        h, w = int(rng.choice([1, 2, 5, 9, 17])), int(rng.choice([1, 3, 7, 20, 63, 64, 65, 70, 130]))
        cin, cout = 64 * int(rng.randint(1, 4)), int(rng.choice([8, 24, 64, 72, 136]))
        x = synth_tensor(60 + case, "fx", (n, cin, t, h, w), -1, 1).to(H).float().requires_grad_()
        wt = (synth_tensor(60 + case, "fw", (cout, cin, 1, 3, 3), -1, 1) * (1.0 / (9 * cin)) ** 0.5).to(H).float().requires_grad_()
        y = F.conv3d(x, wt, padding=(0, 1, 1))
        dy = synth_tensor(60 + case, "fdy", tuple(y.shape), -1, 1).to(H).float()
        y.backward(dy)
        wp = torch.nn.Parameter(wt.detach().clone().cuda())
        layer = TE.ConvLayer(wp, None, (1, 1, 1), (0, 1, 1))
        TE.ARENA.reset("cuda")
        layer.wgrad(cl(x.detach()), cl(dy))
        layer.flush_grad()
        assert rel_l2(wp.grad.cpu(), wt.grad) < 1e-3, (case, n, t, h, w, cin, cout)


@pytest.mark.parametrize("C", [64, 256, 512, 1024, 2048, 2560])
def test_bn_bwd_apply_bias_gradient_covers_every_channel(C):
    """tedspad_bn_bwd_apply's `dbias` rows must add up to the plain per-channel sum of the dz it writes, for every channel -- including
    C >= 512, where one workgroup's 256 threads own more than 256 (chunk, element) pairs (round-2 advisor finding: channels >= 256 were
    dropped; the true bias gradient in front of a train-mode BatchNorm is ~0, so network-level tests could not see it). dz here is made
    non-centred on purpose: sums[] are NOT the real channel sums, so sum(dz) is far from 0."""
    import ctypes as Ct
    from ted_spad_amd import _lib
    L = _lib.lib()
    px, slots = 777, 4
    g = torch.Generator().manual_seed(C)
    dy = torch.randn(px, C, generator=g).to(H).cuda()
    z = torch.randn(px, C, generator=g).cuda()
    mean = (0.1 * torch.randn(C, generator=g)).cuda()
    invstd = (1.0 + 0.1 * torch.rand(C, generator=g)).cuda()
    gamma = (1.0 + 0.2 * torch.randn(C, generator=g)).cuda()
    beta = torch.zeros(C).cuda()
    sums = torch.randn(2, C, generator=g).cuda() * 3.0
    dz = torch.empty(px, C, dtype=H, device="cuda")
    dbias = torch.zeros(slots, C, device="cuda")
    stream = Ct.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = L.tedspad_bn_bwd_apply(dy.data_ptr(), None, z.data_ptr(), _lib.F32, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                sums.data_ptr(), C, dz.data_ptr(), None, dbias.data_ptr(), slots, px, C, C, C, C, C, 0, 0, 1, _lib.F16, stream)
    assert rc == 0, _lib.last_error()
    torch.cuda.synchronize()
    # the kernel's own formula in fp64 (include/tedspad_hip.h): dz = gamma*invstd*(g - sums[0]/M - xhat*sums[1]/M)
    gd, zd = dy.double().cpu(), z.double().cpu()
    ks = (gamma * invstd).double().cpu()
    xhat = (zd - mean.double().cpu()) * invstd.double().cpu()
    ref = ks * (gd - sums[0].double().cpu() / px - xhat * sums[1].double().cpu() / px)
    assert rel_l2(dz.float().cpu().numpy(), ref.numpy()) < 2e-3
    got = dbias.sum(0).double().cpu()
    want = ref.sum(0)
    assert float((got - want).abs().max() / want.abs().max()) < 2e-3, "dbias misses channels: worst %s" % int((got - want).abs().argmax())
    assert int((got == 0).sum()) == 0


def test_bn_bwd_apply_rejects_channel_counts_beyond_its_lds():
    import ctypes as Ct
    from ted_spad_amd import _lib
    t = torch.zeros(8, 4096, device="cuda")
    h = t.to(H)
    rc = _lib.lib().tedspad_bn_bwd_apply(h.data_ptr(), None, t.data_ptr(), _lib.F32, t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), 4096,
                                         h.data_ptr(), None, None, 1, 8, 4096, 4096, 4096, 4096, 4096, 0, 0, 1, _lib.F16,
                                         Ct.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc != 0


@pytest.mark.parametrize("B,K,N", [(2, 2048, 512), (12, 2048, 2048), (16, 2048, 128), (24, 2048, 102), (2048, 12, 2048), (128, 24, 130), (512, 7, 64), (5, 130, 33), (3, 2050, 9)])
def test_linear_kernels_vs_torch(B, K, N):
    """head.linear (tedspad_linear_fwd: Linear (+ folded scale / shift) (+ ReLU), exact fp32) on its three kernels -- a wavefront per output, a wavefront per output
    column for small batches (the mlp heads, model_loaders.py:250-254), a thread per output column for short reductions (their weight gradients dW = dY^T . X) --
    against torch on the CPU (float64 accumulation): forward heads, the transposed products of the backward, K not a multiple of 4."""
    from ted_spad_amd import head
    g = torch.Generator().manual_seed(B * 131 + K)
    x, w, b = torch.randn(B, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
    ref = (x.double() @ w.double().t() + b.double())
    got = head.linear(x.cuda(), w.cuda(), b.cuda()).cpu()
    assert got.shape == (B, N)
    assert rel_l2(got, ref.float()) < 2e-6
    got_r = head.linear(x.cuda(), w.cuda(), b.cuda(), relu=True).cpu()
    assert rel_l2(got_r, ref.clamp_min(0).float()) < 2e-6
