"""-m gpu: each HIP kernel, called through the C ABI, against the CPU oracle on the same
seeded inputs (values pre-rounded to the 16-bit storage type, so the only differences are
fp32 summation order and the final 16-bit rounding)."""
import numpy as np
import pytest
import torch

from conftest import rel_l2
from ted_spad_amd.synth import synth_tensor

pytestmark = pytest.mark.gpu

CASES = [
    # name, (n,t,h,w), cin, cout, k, stride, pads_front, pads_back, residual
    ("pointwise_64_256_res", (2, 4, 13, 11), 64, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), (0, 0, 0), True),
    ("t3x1x1", (2, 4, 9, 10), 256, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 0, 0), False),
    ("s1x3x3", (1, 4, 15, 14), 64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (0, 1, 1), False),
    ("s1x3x3_stride2", (2, 2, 55, 55), 128, 128, (1, 3, 3), (1, 2, 2), (0, 1, 1), (0, 1, 1), False),
    ("down_stride2", (2, 2, 55, 55), 256, 512, (1, 1, 1), (1, 2, 2), (0, 0, 0), (0, 0, 0), False),
    ("k3x3x3_same", (1, 8, 14, 14), 96, 208, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), False),
    ("k3x3x3_cin16", (1, 4, 14, 14), 16, 48, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), False),
    ("k3x3x3_cin24_cout24", (2, 4, 7, 7), 24, 24, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), False),
    ("asym_same_pad", (1, 5, 9, 9), 32, 64, (3, 3, 3), (2, 2, 2), (0, 1, 1), (1, 1, 1), False),
    ("big_k_small_m", (2, 2, 7, 7), 2048, 512, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 0, 0), False),
    ("unet_2d_3x3", (3, 1, 28, 28), 256, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1), (0, 1, 1), False),
    ("ragged_m_1px", (1, 1, 1, 1), 64, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), (0, 0, 0), False),
]


def _round(t, dt):
    return t.to(dt).float()


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_fused(case, dtype):
    from oracle.conv_ref import conv_cl
    from ted_spad_amd import engine as E
    name, dims, cin, cout, k, stride, pf, pb, use_res = case
    tdt = E.DTYPES[dtype][0]
    n, t, h, w = dims
    x = _round(synth_tensor(1, name + "x", (n, t, h, w, cin), -1, 1), tdt)
    wgt = _round(synth_tensor(1, name + "w", (cout, cin) + k, -1, 1) * (2.0 / (cin * k[0] * k[1] * k[2])) ** 0.5, tdt)
    scale = synth_tensor(1, name + "s", (cout,), 0.5, 1.5)
    shift = synth_tensor(1, name + "b", (cout,), -0.3, 0.3)
    ref_nores = conv_cl(x, wgt, scale, shift, stride, pf, pb, None, relu=False)
    res = _round(synth_tensor(1, name + "r", tuple(ref_nores.shape), -1, 1), tdt) if use_res else None
    ref = conv_cl(x, wgt, scale, shift, stride, pf, pb, res, relu=True)
    pc = E.PackedConv(wgt, scale, shift, stride=stride, dtype=dtype, device="cuda")
    xa = E.Act(x.to(tdt).cuda(), cin)
    ra = E.Act(res.to(tdt).cuda(), cout) if use_res else None
    out = pc(xa, pads=pf, pads_back=pb, residual=ra, relu=True)
    torch.cuda.synchronize()
    got = out.buf.float().cpu()
    assert got.shape == ref.shape
    tol = 2.0 ** -10 if dtype == "f16" else 2.0 ** -7
    err = (got - ref).abs()
    bound = tol * ref.abs() + 1e-3
    assert bool((err <= bound).all()), "max err %g at ref %g" % (float(err.max()), float(ref.abs().max()))
    assert rel_l2(got, ref) < (4e-4 if dtype == "f16" else 3e-3)


def test_conv_into_concat_slice_and_from_slice():
    """Inception-style: write into a channel slice of a wider tensor, read from a slice."""
    from oracle.conv_ref import conv_cl
    from ted_spad_amd import engine as E
    tdt = torch.float16
    x = _round(synth_tensor(2, "cx", (2, 4, 7, 7, 96), -1, 1), tdt)
    wgt = _round(synth_tensor(2, "cw", (32, 64, 1, 1, 1), -0.2, 0.2), tdt)
    scale, shift = torch.ones(32), torch.zeros(32)
    ref = conv_cl(x[..., 32:96], wgt, scale, shift, relu=False)
    pc = E.PackedConv(wgt, scale, shift, dtype="f16", device="cuda")
    big = E.Act(torch.full((2, 4, 7, 7, 80), 7.0, dtype=tdt, device="cuda"), 80)
    pc(E.Act(x.to(tdt).cuda(), 96).slice(32, 64), out=big.slice(40, 32), relu=False)
    torch.cuda.synchronize()
    got = big.buf.float().cpu()
    assert bool((got[..., :40] == 7).all()) and bool((got[..., 72:] == 7).all())  # neighbours untouched
    assert rel_l2(got[..., 40:72], ref) < 4e-4


@pytest.mark.parametrize("arch,k,pw", [("largei3d", (5, 7, 7), 3), ("i3d", (7, 7, 7), 2)])
def test_stem_pixel_pair_rewrite(arch, k, pw):
    """Cin=3 stride-2 stems run as an 8-channel conv over pixel pairs (engine.PackedConv pair_w)."""
    from oracle.conv_ref import conv_cl
    from ted_spad_amd import engine as E
    tdt = torch.float16
    clip = _round(synth_tensor(3, "clip" + arch, (2, 3, 8, 32, 32)), tdt)
    wgt = _round(synth_tensor(3, "stemw" + arch, (64, 3) + k, -0.1, 0.1), tdt)
    scale = synth_tensor(3, "stems", (64,), 0.5, 1.5)
    shift = synth_tensor(3, "stemb", (64,), -0.3, 0.3)
    if arch == "largei3d":
        pf, pb = (2, 3, 3), (2, 3, 3)
    else:  # TF-SAME for even sizes, k=7, s=2: (2,3)
        pf, pb = (2, 2, 2), (3, 3, 3)
    ref = conv_cl(clip.permute(0, 2, 3, 4, 1), wgt, scale, shift, (2, 2, 2), pf, pb)
    pc = E.PackedConv(wgt, scale, shift, stride=(2, 2, 2), dtype="f16", device="cuda", pair_w=pw)
    a = E.clip_to_act(clip.cuda(), cpad=4, dtype="f16")
    kw2 = pc.k[2]
    out = pc(a, pads=(pf[0], pf[1], pc.pair_pw), pads_back=(pb[0], pb[1], kw2 - 1 - pc.pair_pw))
    torch.cuda.synchronize()
    got = out.buf.float().cpu()
    assert got.shape == ref.shape
    assert rel_l2(got, ref) < 4e-4


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("variant", [0, 2])
@pytest.mark.parametrize("shape", [(2, 3, 16, 32, 32), (1, 3, 16, 120, 88), (3, 3, 5, 18, 72), (1, 2, 8, 66, 24), (1, 3, 16, 224, 224), (40, 3, 16, 40, 40),
                                   (30, 3, 8, 24, 160)])
def test_stem_persistent_with_temporal_pool(shape, variant, dtype):
    """engine.StemPT (csrc/conv_stem_pt.hip): conv1 5x7x7/2 + bn1 + ReLU of large_i3d.py:133-137 with the temporal half of
    maxpool1 (large_i3d.py:138) fused, against the oracle's Conv3d followed by a max over output-frame pairs: ragged patches,
    rows / columns / frames outside the clip, odd frame counts (the unpaired last frame is dropped like MaxPool3d does), more
    patches than workgroups (persistent loop), and -- with the spatial half of the pool -- the pixel-pair stem + max-pool it replaces."""
    from oracle.conv_ref import conv_cl
    from ted_spad_amd import engine as E
    tdt = E.DTYPES[dtype][0]
    n, c, t, h, w = shape
    clip = _round(synth_tensor(5, "ptclip%d" % h, shape), tdt)
    wgt = _round(synth_tensor(5, "tuw", (64, c, 5, 7, 7), -0.1, 0.1), tdt)
    scale, shift = synth_tensor(5, "tus", (64,), 0.5, 1.5), synth_tensor(5, "tub", (64,), -0.3, 0.3)
    full = conv_cl(clip.permute(0, 2, 3, 4, 1), wgt, scale, shift, (2, 2, 2), (2, 3, 3), (2, 3, 3))     # (n, To, ho, wo, 64), ReLU applied
    tp = full.shape[1] // 2
    ref = torch.maximum(full[:, 0:2 * tp:2], full[:, 1:2 * tp:2])
    st = E.StemPT(wgt, scale, shift, stride=(2, 2, 2), pads=(2, 3, 3), dtype=dtype, device="cuda")
    assert st.applies(clip.cuda())
    got_a = st.conv(st.layout(clip.cuda()), variant=variant)
    got = got_a.buf.float().cpu()
    torch.cuda.synchronize()
    assert got.shape == ref.shape
    ulp = 2.0 ** -10 if dtype == "f16" else 2.0 ** -7
    err = (got - ref).abs()
    assert bool((err <= ulp * ref.abs() + 2e-3).all()), "max err %g" % float(err.max())
    assert rel_l2(got, ref) < (4e-4 if dtype == "f16" else 3e-3)
    if c == 3 and full.shape[2] >= 3 and full.shape[3] >= 3:
        pc = E.PackedConv(wgt, scale, shift, stride=(2, 2, 2), dtype=dtype, device="cuda", pair_w=3)
        old = pc(E.clip_to_act(clip.cuda(), cpad=4, dtype=dtype), pads=(2, 3, pc.pair_pw), pads_back=(2, 3, 1))
        old = E.maxpool(old, (2, 3, 3), (2, 2, 2)).buf.float().cpu()
        new = E.maxpool(got_a, (1, 3, 3), (1, 2, 2)).buf.float().cpu()
        assert new.shape == old.shape
        assert bool(((new - old).abs() <= ulp * old.abs() + 1e-3).all())
        # the whole pool inside the stem kernel (column strips, rows carried between patches, the shared column joined by the
        # fix-up launch): the SAME 16-bit values as pooling the stem kernel's own output
        fused = st.conv_pool(st.layout(clip.cuda()), variant=variant).buf.float().cpu()
        assert fused.shape == new.shape
        assert torch.equal(fused, new)
        # ... and on 16x16x32 MFMAs (variant bit 2: two taps per instruction, another fp32 summation order): against the oracle's pooled tensor
        ref_pool = torch.nn.functional.max_pool3d(ref.permute(0, 4, 1, 2, 3), (1, 3, 3), (1, 2, 2)).permute(0, 2, 3, 4, 1)
        f16x = st.conv_pool(st.layout(clip.cuda()), variant=variant | 4).buf.float().cpu()
        assert f16x.shape == ref_pool.shape
        for got_p in (fused, f16x):
            assert bool(((got_p - ref_pool).abs() <= ulp * ref_pool.abs() + 2e-3).all()), "max err %g" % float((got_p - ref_pool).abs().max())
            assert rel_l2(got_p, ref_pool) < (4e-4 if dtype == "f16" else 3e-3)
        assert bool(((f16x - fused).abs() <= ulp * fused.abs() + 1e-3).all())
        # ... and WITHOUT the layout pass: the kernel's own loader reads the fp32 NCTHW clip (tedspad_stem_pt_pool_clip_fwd) -- the same 16-bit values as
        # the 16x16x32 form on the tedspad_clip_to_tp records; also from a non-contiguous view (a T-slice of a longer clip, Q15)
        if w % 4 == 0:
            xg = clip.cuda()
            assert st.direct_applies(xg)
            direct = st.conv_pool_clip(xg).buf.float().cpu()
            assert torch.equal(direct, f16x)
            big = torch.zeros((n, c, t + 8, h, w), device="cuda")
            big[:, :, 4:4 + t] = xg
            view = big[:, :, 4:4 + t]
            assert not view.is_contiguous() and st.direct_applies(view)
            assert torch.equal(st.conv_pool_clip(view).buf.float().cpu(), f16x)


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("dims", [(3, 2, 28, 28), (2, 3, 7, 9), (1, 1, 16, 16), (4, 2, 14, 30), (1, 2, 56, 56)])
def test_bottleneck_tail_128_mid_channels_vs_oracle_and_unfused(dims, dtype, monkeypatch):
    """engine.BneckTail with 128 mid channels (csrc/conv_bneck.hip, conv_bneck_tail128_kernel): conv2 1x3x3 (128 -> 128) + bn2 + ReLU -> conv3
    (128 -> 512) + bn3 + residual + ReLU of layer2's plain bottlenecks (large_i3d.py:69-84) in one launch -- chunk-major stage A, eight-step
    stage B with the conv3 weights streamed two output groups at a time -- against the oracle (the 128-channel tensor rounded where the
    unfused path stores it) and against the two launches it replaces."""
    from oracle.conv_ref import conv_cl
    from ted_spad_amd import engine as E
    tdt = E.DTYPES[dtype][0]
    n, t, h, w = dims
    name = "bt128_%d" % h
    x = _round(synth_tensor(8, name + "x", (n, t, h, w, 128), -1, 1), tdt)
    w2 = _round(synth_tensor(8, name + "w2", (128, 128, 1, 3, 3), -1, 1) * (2.0 / 1152) ** 0.5, tdt)
    w3 = _round(synth_tensor(8, name + "w3", (512, 128, 1, 1, 1), -1, 1) * (2.0 / 128) ** 0.5, tdt)
    s2, b2 = synth_tensor(8, name + "s2", (128,), 0.5, 1.5), synth_tensor(8, name + "b2", (128,), -0.3, 0.3)
    s3, b3 = synth_tensor(8, name + "s3", (512,), 0.5, 1.5), synth_tensor(8, name + "b3", (512,), -0.3, 0.3)
    res = _round(synth_tensor(8, name + "r", (n, t, h, w, 512), -1, 1), tdt)
    mid = _round(conv_cl(x, w2, s2, b2, (1, 1, 1), (0, 1, 1), (0, 1, 1), None, relu=True), tdt)
    ref = conv_cl(mid, w3, s3, b3, (1, 1, 1), (0, 0, 0), (0, 0, 0), res, relu=True)
    c2 = E.PackedConv(w2, s2, b2, dtype=dtype, device="cuda")
    tail = E.BneckTail(c2, w3, s3, b3)
    xa = E.Act(x.to(tdt).cuda(), 128)
    assert tail.cmid == 128 and tail.applies(xa, (0, 1, 1))
    got = tail(xa, residual=E.Act(res.to(tdt).cuda(), 512)).buf.float().cpu()
    assert got.shape == ref.shape
    ulp = 2.0 ** -10 if dtype == "f16" else 2.0 ** -7
    err = (got - ref).abs()
    assert bool((err <= ulp * ref.abs() + 2e-3).all()), "max err %g" % float(err.max())
    assert rel_l2(got, ref) < (4e-4 if dtype == "f16" else 3e-3)
    monkeypatch.setattr(E, "FORCE_TILE_CFG", None)
    h2 = c2(xa, pads=(0, 1, 1))
    old = E.PackedConv(w3, s3, b3, dtype=dtype, device="cuda")(h2, residual=E.Act(res.to(tdt).cuda(), 512), relu=True).buf.float().cpu()
    # the unfused conv2 may run a K-order-preserving tile, the fused stage A walks (chunk, tap): the 16-bit intermediate can differ by one step
    assert rel_l2(got, old) < (6e-4 if dtype == "f16" else 4e-3)


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("dims", [(3, 14, 14, 1024, 256), (5, 7, 7, 2048, 512), (2, 28, 28, 512, 256), (1, 5, 3, 128, 256), (2, 28, 28, 256, 128), (3, 9, 11, 512, 384)])
def test_temporal_conv_on_two_frames_as_one_folded_gemm(dims, dtype, monkeypatch):
    """engine.TPairConv: a 3x1x1 'same' conv + BN + ReLU on a 2-frame tensor (conv1 of layer3 / layer4's temporal bottlenecks,
    large_i3d.py:61-68 with T = 2 behind maxpool2) as ONE K = 2*cin GEMM over both frames whose 2*cout output channels are the two
    output frames (tedspad_conv_extras.fold_hw on the ping-pong kernel) -- against the oracle's Conv3d and against the K = 3*cin launch it
    replaces (same products minus the ones on zero padding: one f16 rounding step apart at most)."""
    from oracle.conv_ref import conv_cl
    from ted_spad_amd import engine as E
    tdt = E.DTYPES[dtype][0]
    n, h, w, cin, cout = dims
    name = "tp%d%d" % (h, cin)
    x = _round(synth_tensor(9, name + "x", (n, 2, h, w, cin), -1, 1), tdt)
    wgt = _round(synth_tensor(9, name + "w", (cout, cin, 3, 1, 1), -1, 1) * (2.0 / (3 * cin)) ** 0.5, tdt)
    scale, shift = synth_tensor(9, name + "s", (cout,), 0.5, 1.5), synth_tensor(9, name + "b", (cout,), -0.3, 0.3)
    ref = conv_cl(x, wgt, scale, shift, (1, 1, 1), (1, 0, 0), (1, 0, 0), None, relu=True)
    tp = E.TPairConv(wgt, scale, shift, dtype=dtype, device="cuda")
    xa = E.Act(x.to(tdt).cuda(), cin)
    assert tp.applies(xa, (1, 0, 0)) and not tp.applies(xa, (0, 0, 0))
    for cfg in (25, 26, None):                      # both MFMA shapes of the ping-pong kernel, then the tuner's own pick
        monkeypatch.setattr(E, "FORCE_TILE_CFG", cfg)
        got = tp(xa).buf.float().cpu()
        assert got.shape == ref.shape
        ulp = 2.0 ** -10 if dtype == "f16" else 2.0 ** -7
        err = (got - ref).abs()
        assert bool((err <= ulp * ref.abs() + 2e-3).all()), "cfg %s: max err %g" % (cfg, float(err.max()))
        assert rel_l2(got, ref) < (4e-4 if dtype == "f16" else 3e-3)
    monkeypatch.setattr(E, "FORCE_TILE_CFG", None)
    old = E.PackedConv(wgt, scale, shift, dtype=dtype, device="cuda")(xa, pads=(1, 0, 0)).buf.float().cpu()
    assert bool(((got - old).abs() <= ulp * old.abs() + 1e-3).all())
    monkeypatch.setattr(E, "FORCE_TILE_CFG", 22)    # any other tile refuses the folded epilogue
    with pytest.raises(RuntimeError):
        tp(xa)


@pytest.mark.parametrize("shape,lo,frames", [((2, 3, 48, 10, 70), 16, 16), ((2, 3, 40, 6, 72), 4, 32), ((1, 3, 16, 5, 224), 0, 16)])
def test_clip_to_frame_pair_layout(shape, lo, frames):
    """tedspad_clip_to_tp: record (tp, h, b, wq) value dt*3 + c = x[n][c][4*tp - 2 + dt][h][2*wq + b], zeros outside the clip;
    strided (Q15) input views, a ragged last tile. W = 70: the four-byte-load kernel; W = 72 / 224 (16-byte aligned rows): the float4 kernel, with 32 frames two
    groups of four pairs."""
    from ted_spad_amd import engine as E
    big = synth_tensor(6, "tcbig%d" % shape[-1], shape)
    x = big[:, :, lo:lo + frames]                                 # a torch.split-style view: not contiguous
    n, _, _, h, w = shape
    tp_n = frames // 4
    st = E.StemPT(torch.zeros(64, 3, 5, 7, 7), None, None, dtype="f16", device="cuda")
    got = st.layout(x.cuda()).float().cpu()                       # (n, tp, h, 2, w / 2, 24)
    xp = torch.zeros(n, 3, 2 + frames + 4, h, w)
    xp[:, :, 2:2 + frames] = x.half().float()
    ref = torch.stack([xp[:, :, 4 * tp:4 * tp + 8] for tp in range(tp_n)], dim=1)      # (n, tp, c, dt, h, w)
    ref = ref.permute(0, 1, 4, 5, 3, 2).reshape(n, tp_n, h, w // 2, 2, 24).permute(0, 1, 2, 4, 3, 5)
    assert torch.equal(got, ref)


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("dual", [False, True])
@pytest.mark.parametrize("dims", [(3, 4, 55, 55), (2, 3, 7, 9), (1, 1, 16, 16), (5, 2, 28, 30), (3, 6, 9, 11)])
def test_bottleneck_tail_fused_vs_oracle_and_unfused(dims, dual, dtype):
    """engine.BneckTail (csrc/conv_bneck.hip): conv2 1x3x3 (64 -> 64) + bn2 + ReLU -> conv3 1x1x1 (64 -> 256) + bn3 + (residual | downsample
    branch) + ReLU of a layer1 bottleneck (large_i3d.py:69-84) in one launch, against the oracle (the 64-channel tensor rounded to the
    storage type where the unfused path stores it) and against the two unfused launches it replaces. Tiles crossing rows, frames and
    clips, a ragged last tile, frames smaller than a tile."""
    from oracle.conv_ref import conv_cl
    from ted_spad_amd import engine as E
    tdt = E.DTYPES[dtype][0]
    n, t, h, w = dims
    name = "bt%d%d" % (h, int(dual))
    x = _round(synth_tensor(8, name + "x", (n, t, h, w, 64), -1, 1), tdt)
    w2 = _round(synth_tensor(8, name + "w2", (64, 64, 1, 3, 3), -1, 1) * (2.0 / 576) ** 0.5, tdt)
    w3 = _round(synth_tensor(8, name + "w3", (256, 64, 1, 1, 1), -1, 1) * (2.0 / 64) ** 0.5, tdt)
    wd = _round(synth_tensor(8, name + "wd", (256, 64, 1, 1, 1), -1, 1) * (2.0 / 64) ** 0.5, tdt)
    s2, b2 = synth_tensor(8, name + "s2", (64,), 0.5, 1.5), synth_tensor(8, name + "b2", (64,), -0.3, 0.3)
    s3, b3 = synth_tensor(8, name + "s3", (256,), 0.5, 1.5), synth_tensor(8, name + "b3", (256,), -0.3, 0.3)
    sd, bd = synth_tensor(8, name + "sd", (256,), 0.5, 1.5), synth_tensor(8, name + "bd", (256,), -0.3, 0.3)
    res = _round(synth_tensor(8, name + "r", (n, t, h, w, 256), -1, 1), tdt)
    x2 = _round(synth_tensor(8, name + "x2", (n, t, h, w, 64), -1, 1), tdt)
    mid = _round(conv_cl(x, w2, s2, b2, (1, 1, 1), (0, 1, 1), (0, 1, 1), None, relu=True), tdt)
    if dual:
        ref = torch.relu(conv_cl(mid, w3, s3, b3, relu=False) + conv_cl(x2, wd, sd, bd, relu=False))
    else:
        ref = conv_cl(mid, w3, s3, b3, (1, 1, 1), (0, 0, 0), (0, 0, 0), res, relu=True)
    c2 = E.PackedConv(w2, s2, b2, dtype=dtype, device="cuda")
    tail = E.BneckTail(c2, w3, s3, b3, wd if dual else None, sd if dual else None, bd if dual else None)
    xa = E.Act(x.to(tdt).cuda(), 64)
    assert tail.applies(xa, (0, 1, 1))
    if dual:
        got = tail(xa, x2=E.Act(x2.to(tdt).cuda(), 64))
    else:
        got = tail(xa, residual=E.Act(res.to(tdt).cuda(), 256))
    torch.cuda.synchronize()
    got = got.buf.float().cpu()
    assert got.shape == ref.shape
    ulp = 2.0 ** -10 if dtype == "f16" else 2.0 ** -7
    err = (got - ref).abs()
    assert bool((err <= ulp * ref.abs() + 2e-3).all()), "max err %g" % float(err.max())
    assert rel_l2(got, ref) < (4e-4 if dtype == "f16" else 3e-3)
    # the unfused launches: identical 64-channel intermediate, conv3 sums its 64 products in another order (fp32): one rounding step apart at most
    E.FORCE_TILE_CFG = None
    h2 = c2(xa, pads=(0, 1, 1))
    c3 = E.PackedConv(w3, s3, b3, dtype=dtype, device="cuda")
    if dual:
        cd = E.PackedConv(wd, sd, bd, dtype=dtype, device="cuda")
        old = c3.call_dual(h2, cd, E.Act(x2.to(tdt).cuda(), 64), relu=True)
    else:
        old = c3(h2, residual=E.Act(res.to(tdt).cuda(), 256), relu=True)
    old = old.buf.float().cpu()
    assert bool(((got - old).abs() <= ulp * old.abs() + 1e-3).all())
    if not dual and t % 2 == 0:
        # maxpool2 (MaxPool3d((2,1,1), stride (2,1,1)), large_i3d.py:139) fused as well: the SAME values as pooling the fused output
        pooled = tail(xa, residual=E.Act(res.to(tdt).cuda(), 256), pool_t2=True).buf.float().cpu()
        want = torch.maximum(got[:, 0::2], got[:, 1::2])
        assert pooled.shape == want.shape
        assert torch.equal(pooled, want)


POOLS = [
    ("res_maxpool1", (2, 8, 30, 30), 64, (2, 3, 3), (2, 2, 2), (0, 0, 0), (0, 0, 0), False),
    ("res_maxpool2", (2, 4, 9, 9), 256, (2, 1, 1), (2, 1, 1), (0, 0, 0), (0, 0, 0), False),
    ("inc_1x3x3_s2_same", (1, 8, 28, 28), 64, (1, 3, 3), (1, 2, 2), (0, 0, 0), (0, 1, 1), True),
    ("inc_3x3x3_s1_same", (1, 4, 14, 14), 480, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), True),
    ("inc_3x3x3_s1_odd_hw", (2, 2, 7, 7), 832, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), True),
    ("k3s1_skip_padding", (2, 3, 5, 6), 16, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), False),
    ("k3s1_degenerate", (1, 1, 1, 1), 8, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), True),
    ("k3s1_banded_28", (2, 3, 28, 28), 200, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), True),        # LDS frame tiles: four row bands with halo rows, a ragged last channel group (25 = 3 x 8 + 1)
    ("k3s1_banded_odd", (1, 5, 19, 23), 72, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), False),      # ... odd frame, a short last band, -inf padding
    ("k3s1_wide_column_walk", (1, 2, 5, 40), 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), True),  # W x 8 groups > 256 threads: the column-walking kernel
    ("inc_3x3x3_s2_same", (1, 8, 28, 28), 32, (3, 3, 3), (2, 2, 2), (0, 0, 0), (1, 1, 1), True),
    ("inc_2x2x2_s2_odd", (1, 4, 7, 7), 64, (2, 2, 2), (2, 2, 2), (0, 0, 0), (0, 1, 1), True),
    ("unet_2x2", (3, 1, 14, 14), 128, (1, 2, 2), (1, 2, 2), (0, 0, 0), (0, 0, 0), False),
]


@pytest.mark.parametrize("case", POOLS, ids=[c[0] for c in POOLS])
def test_maxpool_bit_exact(case):
    from oracle.conv_ref import maxpool_cl
    from ted_spad_amd import engine as E
    name, dims, c, k, s, pf, pb, pz = case
    # signed inputs: zero-padding semantics (i3d.py:41-45) differ from -inf padding here
    x = synth_tensor(4, name, dims + (c,), -1, 1).half()
    ref = maxpool_cl(x.float(), k, s, pf, pb, pad_zero=pz)
    out = E.maxpool(E.Act(x.cuda(), c), k, s, pf, pb, pad_zero=pz)
    torch.cuda.synchronize()
    assert torch.equal(out.buf.float().cpu(), ref)


def test_maxpool_k3s1_bf16_bit_exact():
    """The 3x3x3 / stride 1 fast path in bf16 storage (max goes through fp32 there)."""
    from oracle.conv_ref import maxpool_cl
    from ted_spad_amd import engine as E
    x = synth_tensor(4, "k3s1bf", (2, 4, 9, 10, 24), -1, 1).bfloat16()
    for pz in (True, False):
        ref = maxpool_cl(x.float(), (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), pad_zero=pz)
        out = E.maxpool(E.Act(x.cuda(), 24), (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), pad_zero=pz)
        assert torch.equal(out.buf.float().cpu(), ref)


def test_avgpool_and_layouts():
    from ted_spad_amd import engine as E
    x = synth_tensor(5, "avg", (3, 2, 7, 7, 2048), 0, 4).half()
    got = E.global_avgpool(E.Act(x.cuda(), 2048)).cpu()
    ref = x.float().mean(dim=(1, 2, 3))
    assert rel_l2(got, ref) < 1e-6
    clip = synth_tensor(5, "clip", (2, 3, 4, 6, 8))
    for cpad in (4, 8):
        a = E.clip_to_act(clip.cuda().permute(0, 1, 2, 4, 3).contiguous().permute(0, 1, 2, 4, 3), cpad=cpad)  # strided input
        cl = a.buf.float().cpu().reshape(2, 4, 6, 8, cpad)
        assert torch.equal(cl[..., :3], clip.permute(0, 2, 3, 4, 1).half().float())
        assert bool((cl[..., 3:] == 0).all())
    back = E.act_to_nchw(E.Act(clip.permute(0, 2, 3, 4, 1).half().cuda().contiguous(), 3), 3)
    assert torch.equal(back.cpu(), clip.half().float())


def test_bad_arguments_fail_loudly():
    from ted_spad_amd import _lib, engine as E
    with pytest.raises(_lib.TedSpadHipError):
        E.clip_to_act(torch.zeros(1, 3, 2, 4, 4), cpad=4)  # CPU tensor: no CPU path
    pc = E.PackedConv(torch.zeros(8, 8, 1, 1, 1), torch.ones(8), torch.zeros(8), device="cuda")
    with pytest.raises(AssertionError):
        pc(E.Act(torch.zeros(1, 1, 2, 2, 16, dtype=torch.float16, device="cuda"), 16))


AGREE = [
    ("a_1x3x3_c64", (3, 2, 19, 23), 64, 64, (1, 3, 3), (0, 1, 1), True),
    ("a_3x1x1_c128", (2, 4, 9, 11), 128, 256, (3, 1, 1), (1, 0, 0), False),
    ("a_1x1x1_c256", (5, 2, 7, 9), 256, 128, (1, 1, 1), (0, 0, 0), True),
    ("a_3x3_cin24", (2, 1, 21, 17), 24, 40, (1, 3, 3), (0, 1, 1), False),
    ("a_1x1x1_c64_res", (7, 3, 13, 11), 64, 256, (1, 1, 1), (0, 0, 0), True),        # persistent pointwise (19): ragged M
    ("a_1x1x1_c128", (3, 2, 9, 10), 128, 72, (1, 1, 1), (0, 0, 0), False),            # ... cin 128, ragged N
    ("a_1x1x1_c64_big", (40, 2, 28, 28), 64, 64, (1, 1, 1), (0, 0, 0), True),         # ... many tiles per persistent workgroup
    ("a_flat_3x3_c64_w55", (2, 3, 11, 55), 64, 64, (1, 3, 3), (0, 1, 1), False),     # flat-halo tile (27): tiles cross rows, frames and clips
    ("a_flat_3x3_c64_cout40", (1, 2, 9, 7), 64, 40, (1, 3, 3), (0, 1, 1), True),     # ... frames smaller than a tile, ragged N, residual
    ("a_tflat_3x1x1_c256", (3, 4, 9, 11), 256, 64, (3, 1, 1), (1, 0, 0), False),      # temporal flat-halo tile (28): T = 4, ragged spatial tile
    ("a_tflat_3x1x1_t3_c64", (2, 3, 5, 13), 64, 48, (3, 1, 1), (1, 0, 0), True),      # ... T = 3 (an idle wave), one chunk, ragged N, residual
    ("a_cflat_3x3_c128_n128", (3, 2, 13, 28), 128, 128, (1, 3, 3), (0, 1, 1), True),   # flat chunk-major tile (33): tiles cross rows / frames / clips
    ("a_temp_3x1x1_t2_c128_n256", (3, 2, 9, 15), 128, 256, (3, 1, 1), (1, 0, 0), True),  # temporal chunk-major tile (34): T = 2 (a third of the taps skipped), two channel tiles
    ("a_temp_3x1x1_t3_c64_n72", (2, 3, 5, 7), 64, 72, (3, 1, 1), (1, 0, 0), False),      # ... T = 3 (192 of 256 positions), ragged N
    ("a_p2_3x3_c32_n32", (2, 1, 21, 37), 32, 32, (1, 3, 3), (0, 1, 1), True),             # two-patch tile (38): one half chunk, two of four channel groups, odd patch count
    ("a_p2_3x3_c96_n8", (1, 2, 18, 20), 96, 8, (1, 3, 3), (0, 1, 1), False),              # ... three half chunks, one channel group (the unet++ head)
    ("a_p2_3x3_c128_n136", (3, 1, 33, 16), 128, 136, (1, 3, 3), (0, 1, 1), True),          # ... three channel tiles, the last with one group
    ("a_patch_3x3_c128_n320", (2, 2, 14, 14), 128, 320, (1, 3, 3), (0, 1, 1), True),   # patch / flat chunk-major tiles with three channel tiles (N-tiling)
    ("a_patch_3x3_c64", (2, 1, 20, 37), 64, 64, (1, 3, 3), (0, 1, 1), True),          # patch-halo tile (32): ragged 16 x 16 patches, residual
    ("a_patch_3x3_c128_n128", (1, 2, 17, 16), 128, 128, (1, 3, 3), (0, 1, 1), False),  # ... two channel chunks, 128 output channels (2-slot ring)
    ("a_patch_3x3_c192_n72", (1, 1, 9, 33), 192, 72, (1, 3, 3), (0, 1, 1), False),     # ... three chunks, ragged N in the second staging pass
    ("a_p8_1x3x3_c256", (3, 2, 14, 13), 256, 256, (1, 3, 3), (0, 1, 1), False),       # ping-pong tile (25): ragged M, 36 K tiles
    ("a_p8_3x1x1_c128_res", (2, 4, 9, 11), 128, 512, (3, 1, 1), (1, 0, 0), True),     # ... two channel tiles, residual, 6 K tiles
    ("a_p8_1x1x1_k128", (5, 2, 17, 9), 128, 256, (1, 1, 1), (0, 0, 0), True),         # ... the shortest K it takes (2 K tiles)
    ("a_p8_1x1x1_k192", (1, 1, 5, 7), 192, 256, (1, 1, 1), (0, 0, 0), False),         # ... odd number of K tiles, one ragged pixel tile
    ("a_pw_1x1x1_c256_res", (3, 2, 13, 11), 256, 1024, (1, 1, 1), (0, 0, 0), True),   # layer3's conv3 shape (cin 256 -> 1024 + residual): ragged M, every generic tile
    ("a_patch_3x3x3_c64_n192", (2, 4, 20, 19), 64, 192, (3, 3, 3), (1, 1, 1), False),   # patch tile (32) with temporal taps as chunks: InceptionI3d's Conv3d_2c_3x3 (64 -> 192), ragged patches
    ("a_patch_3x3x3_c128_t2", (3, 2, 9, 17), 128, 96, (3, 3, 3), (1, 1, 1), True),      # ... two channel chunks per temporal tap, T = 2 (every frame skips a tap), ragged N, residual
    ("a_patch_3x3x3_c64_t1", (2, 1, 16, 16), 64, 64, (3, 3, 3), (1, 1, 1), False),      # ... a single frame: only the centre temporal tap runs
    ("a_cflat_3x3x3_c64_n192", (3, 4, 9, 13), 64, 192, (3, 3, 3), (1, 1, 1), True),     # flat tile (33) with temporal taps: tiles span frames AND clips (per-pixel frame validity)
    ("a_igemm_3x3x3_c96_n208", (1, 4, 9, 10), 96, 208, (3, 3, 3), (1, 1, 1), True),      # InceptionI3d's Mixed_4b b1b (96 -> 208): 128-wide tile + its 64-wide sibling on the last 80 -> here 128 + 80: unsplit (r > 64)
    ("a_igemm_3x3x3_c144_n288", (1, 2, 7, 9), 144, 288, (3, 3, 3), (1, 1, 1), False),    # ... Mixed_4e b1b (144 -> 288 = 2 x 128 + 32): the split generic tiles (round 5), K table, ragged everything
    ("a_1x1x1_c192_n176", (2, 4, 9, 10), 192, 176, (1, 1, 1), (0, 0, 0), True),          # ... a fused reduce GEMM (b1a | b2a | b0 = 96 + 16 + 64): 128 + 48 channels, residual
]


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("case", AGREE, ids=[c[0] for c in AGREE])
def test_every_tile_configuration_gives_the_same_result(case, dtype):
    """The tile tuner may pick any applicable configuration, so the choice must not change results: every generic
    implicit-GEMM configuration accumulates K in the same order and must agree BIT-EXACTLY with the others; the
    halo-direct ones (15, 16) walk K as (channel chunk, tap) instead of (tap, channel chunk) -- an fp32 reassociation --
    and must stay within one f16 rounding step of them (and inside the oracle bound of test_conv_fused)."""
    from oracle.conv_ref import conv_cl
    from ted_spad_amd import _lib, engine as E
    name, dims, cin, cout, k, pf, use_res = case
    n, t, h, w = dims
    tdt = E.DTYPES[dtype][0]
    ulp = 2.0 ** -10 if dtype == "f16" else 2.0 ** -7
    x = synth_tensor(7, name + "x", (n, t, h, w, cin), -1, 1).to(tdt).float()
    wgt = (synth_tensor(7, name + "w", (cout, cin) + k, -1, 1) * (2.0 / (cin * k[0] * k[1] * k[2])) ** 0.5).to(tdt).float()
    scale, shift = synth_tensor(7, name + "s", (cout,), 0.5, 1.5), synth_tensor(7, name + "b", (cout,), -0.3, 0.3)
    res = synth_tensor(7, name + "r", (n, t, h, w, cout), -1, 1).to(tdt).float() if use_res else None
    ref = conv_cl(x, wgt, scale, shift, (1, 1, 1), pf, pf, res, relu=True)
    pc = E.PackedConv(wgt, scale, shift, dtype=dtype, device="cuda")
    xa, ra = E.Act(x.to(tdt).cuda(), cin), (E.Act(res.to(tdt).cuda(), cout) if use_res else None)
    outs = {}
    try:
        for cfg in range(1, _lib.lib().tedspad_conv_num_tile_cfgs() + 1):
            E.FORCE_TILE_CFG = cfg
            try:
                outs[cfg] = pc(xa, pads=pf, residual=ra, relu=True).buf.float().cpu()
            except _lib.TedSpadHipError:
                continue                                   # configuration not applicable to this geometry
    finally:
        E.FORCE_TILE_CFG = None
    REASSOC = (15, 16, 22, 23, 24, 26, 28, 32, 33, 34, 35, 36, 37, 38, 39, 40)  # halo-direct (K walked chunk-major), split-K tiles, 16x16x32 MFMA: fp32 sums re-associated
    generic = {c: o for c, o in outs.items() if c not in REASSOC}
    assert len(generic) >= 4, sorted(outs)
    first = next(iter(generic.values()))
    for c, o in generic.items():
        assert torch.equal(o, first), "configuration %d differs from configuration %d" % (c, next(iter(generic)))
    assert bool(((first - ref).abs() <= ulp * ref.abs() + 1e-3).all())
    for c in REASSOC:
        if c in outs:
            assert bool(((outs[c] - first).abs() <= ulp * first.abs() + 1e-4).all()), c
    if 40 in outs:       # the persistent two-patch tile walks K exactly as the two-patch tile does: same bits
        assert 38 in outs and torch.equal(outs[40], outs[38])
    print(name, "configurations run:", sorted(outs))


@pytest.mark.parametrize("dims,cout,ld2", [((3, 4, 11, 13), 256, 64), ((1, 2, 30, 31), 256, 128), ((2, 1, 5, 5), 64, 64)])
def test_dual_pointwise_equals_two_convs(dims, cout, ld2):
    """tedspad_conv_pw_dual_fwd (conv3 + bn3 and the downsample branch of layer1.0 in one launch) against the oracle's two
    convolutions summed in fp32, and against the two-launch path it replaces (which rounds the downsample branch to 16 bits)."""
    from oracle.conv_ref import conv_cl
    from ted_spad_amd import engine as E
    n, t, h, w = dims
    tdt = torch.float16
    x = synth_tensor(11, "dx", (n, t, h, w, 64), -1, 1).to(tdt).float()
    x2full = synth_tensor(11, "dx2", (n, t, h, w, ld2), -1, 1).to(tdt).float()
    x2 = x2full[..., ld2 - 64:]                                   # a channel slice of a wider buffer when ld2 > 64
    w1 = (synth_tensor(11, "dw1", (cout, 64, 1, 1, 1), -1, 1) * (2.0 / 64) ** 0.5).to(tdt).float()
    w2 = (synth_tensor(11, "dw2", (cout, 64, 1, 1, 1), -1, 1) * (2.0 / 64) ** 0.5).to(tdt).float()
    s1, b1 = synth_tensor(11, "ds1", (cout,), 0.5, 1.5), synth_tensor(11, "db1", (cout,), -0.3, 0.3)
    s2, b2 = synth_tensor(11, "ds2", (cout,), 0.5, 1.5), synth_tensor(11, "db2", (cout,), -0.3, 0.3)
    z = (0, 0, 0)
    a = conv_cl(x, w1, s1, b1, (1, 1, 1), z, z, None, relu=False)
    b = conv_cl(x2, w2, s2, b2, (1, 1, 1), z, z, None, relu=False)
    pc1 = E.PackedConv(w1, s1, b1, dtype="f16", device="cuda")
    pc2 = E.PackedConv(w2, s2, b2, dtype="f16", device="cuda")
    xa = E.Act(x.to(tdt).cuda(), 64)
    x2a = E.Act(x2full.to(tdt).cuda(), 64, ld2 - 64)
    assert pc1.dual_supported(pc2, xa, x2a)
    got = pc1.call_dual(xa, pc2, x2a, relu=True).buf.float().cpu()
    two = pc1(xa, residual=pc2(x2a, relu=False), relu=True).buf.float().cpu()
    torch.cuda.synchronize()
    ref = torch.relu(a + b)         # fp32; the fused launch rounds only this sum to f16 (fp32 summation order differs)
    err = (got - ref).abs()
    assert bool((err <= 2.0 ** -10 * (a.abs() + b.abs()) + 1e-3).all()), "max err %g" % float(err.max())
    assert rel_l2(got, ref) < 4e-4 and rel_l2(got, two) < 8e-4


@pytest.mark.parametrize("cin,cout", [(64, 64), (128, 128), (128, 40)])
def test_patch_halo_training_epilogues(cin, cout):
    """tile_cfg 32 with the training extras of the conv epilogue -- ReLU-backward mask (+ residual), batch statistics, fp32 output --
    against a generic tile on the same launch arguments (the extras' own arithmetic is checked in test_hip_train_ops.py)."""
    from ted_spad_amd import engine as E
    tdt = torch.float16
    n, t, h, w = 2, 1, 37, 21
    x = E.Act(synth_tensor(31, "pe_x%d" % cin, (n, t, h, w, cin), -1, 1).to(tdt).cuda(), cin)
    wgt = (synth_tensor(31, "pe_w%d%d" % (cin, cout), (cout, cin, 1, 3, 3), -1, 1) * (2.0 / (9 * cin)) ** 0.5).to(tdt).float()
    pc = E.PackedConv(wgt, synth_tensor(31, "pe_s", (cout,), 0.5, 1.5), synth_tensor(31, "pe_b", (cout,), -0.3, 0.3), dtype="f16", device="cuda")
    res = E.Act(synth_tensor(31, "pe_r", (n, t, h, w, pc.cout), -1, 1).to(tdt).cuda(), pc.cout)
    mask = E.Act(synth_tensor(31, "pe_m", (n, t, h, w, pc.cout), -1, 1).to(tdt).cuda(), pc.cout)
    got = {}
    try:
        for cfg in (32, 33, 38, 40, 5):
            E.FORCE_TILE_CFG = cfg
            if cfg == 40 and not 32 < cout <= 64:
                continue
            st1 = torch.zeros((2, pc.cpad), device="cuda"); st2 = torch.zeros((2, pc.cpad), device="cuda")
            a = pc(x, pads=(0, 1, 1), residual=res, mask=mask, relu=False).buf.float().cpu()
            bq = pc(x, pads=(0, 1, 1), relu=True, stats=st1).buf.float().cpu()
            c32 = pc(x, pads=(0, 1, 1), relu=False, stats=st2, y32=True).cpu()
            got[cfg] = (a, bq, c32, st1.cpu(), st2.cpu())
    finally:
        E.FORCE_TILE_CFG = None
    ulp = 2.0 ** -10
    if 40 in got:               # the persistent tile: tile 38's sums bit for bit (16-bit and fp32 outputs); statistics: float atomics in another order
        for i in range(3):
            assert torch.equal(got[40][i], got[38][i]), i
    for c in (32, 33, 38) + ((40,) if 40 in got else ()):
        for i in range(3):      # cin = 64 walks K like the generic tile (bit-identical), cin = 128 chunk-major (one rounding step; fp32 output: 1e-5); 38 walks half chunks
            g, r = got[c][i], got[5][i]
            assert g.shape == r.shape
            if cin == 64 and c < 38:
                assert torch.equal(g, r), (c, i)
            else:
                assert bool(((g - r).abs() <= (ulp if i < 2 else 2e-5) * r.abs() + 1e-4).all()), (c, i)
        for i in (3, 4):        # batch statistics: float atomics in a different order
            assert rel_l2(got[c][i], got[5][i]) < 1e-5, (c, i)


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("dims,cin,cout,nwg", [((2, 1, 20, 37), 64, 64, 1), ((2, 1, 20, 37), 64, 64, 4), ((3, 1, 33, 16), 128, 64, 2), ((1, 2, 18, 20), 96, 40, 3),
                                              ((5, 1, 48, 48), 192, 64, 7), ((1, 1, 16, 16), 32, 48, 0), ((3, 1, 112, 112), 64, 64, 0), ((2, 1, 50, 70), 320, 64, 5),
                                              ((2, 1, 20, 37), 64, 128, 3), ((2, 1, 33, 48), 192, 128, 0)])       # cout = 128: two 64-channel launches
def test_persistent_two_patch_tile_equals_the_two_patch_tile(dims, cin, cout, nwg, dtype, monkeypatch):
    """tile_cfg 40 (conv_patch3.hip: persistent 8-wave workgroups, double-buffered halo, weight ring -- resident for cin <= 64, streamed otherwise --,
    epilogue from the accumulators) against tile 38, whose K order it keeps: BIT-EQUAL 16-bit outputs with residual + ReLU, with the ReLU-backward mask, and
    fp32 outputs; batch statistics (summed in registers over a workgroup's whole run of tiles) to fp32 rounding; and against the oracle's conv. `nwg` forces a
    small persistent grid (TEDSPAD_P3_NWG) so that a workgroup walks several tiles: odd patch counts, ragged patches, the weight ring wrapping across tiles."""
    from oracle.conv_ref import conv_cl
    from ted_spad_amd import engine as E
    tdt = E.DTYPES[dtype][0]
    n, t, h, w = dims
    name = "p3_%d_%d_%d" % (cin, cout, h)
    if nwg:
        monkeypatch.setenv("TEDSPAD_P3_NWG", str(nwg))
    x = synth_tensor(41, name + "x", (n, t, h, w, cin), -1, 1).to(tdt)
    wgt = (synth_tensor(41, name + "w", (cout, cin, 1, 3, 3), -1, 1) * (2.0 / (9 * cin)) ** 0.5).to(tdt).float()
    scale, shift = synth_tensor(41, name + "s", (cout,), 0.5, 1.5), synth_tensor(41, name + "b", (cout,), -0.3, 0.3)
    pc = E.PackedConv(wgt, scale, shift, dtype=dtype, device="cuda")
    xa = E.Act(x.cuda(), cin)
    res = E.Act(synth_tensor(41, name + "r", (n, t, h, w, pc.cout), -1, 1).to(tdt).cuda(), pc.cout)
    mask = E.Act(synth_tensor(41, name + "m", (n, t, h, w, pc.cout), -1, 1).to(tdt).cuda(), pc.cout)
    got = {}
    try:
        for cfg in (38, 40):
            E.FORCE_TILE_CFG = cfg
            st1 = torch.zeros((2, pc.cpad), device="cuda"); st2 = torch.zeros((2, pc.cpad), device="cuda")
            a = pc(xa, pads=(0, 1, 1), residual=res, relu=True).buf.float().cpu()
            b = pc(xa, pads=(0, 1, 1), residual=res, mask=mask, relu=False).buf.float().cpu()
            c = pc(xa, pads=(0, 1, 1), relu=True, stats=st1).buf.float().cpu()
            d = pc(xa, pads=(0, 1, 1), relu=False, stats=st2, y32=True).cpu()
            got[cfg] = (a, b, c, d, st1.cpu(), st2.cpu())
    finally:
        E.FORCE_TILE_CFG = None
    for i in range(4):
        assert torch.equal(got[40][i], got[38][i]), i
    for i in (4, 5):
        assert rel_l2(got[40][i], got[38][i]) < 1e-5, i
    ref = conv_cl(x.float(), wgt, scale, shift, (1, 1, 1), (0, 1, 1), (0, 1, 1), res.buf.float().cpu()[..., :cout], relu=True)
    assert rel_l2(got[40][0][..., :cout], ref) < (2e-3 if dtype == "f16" else 1.2e-2)


def test_flat_halo_kernels_on_awkward_geometries():
    """The flat-halo tiles (27: 1 x kh x kw, 28: kt x 1 x 1) against a generic tile on seeded random small geometries: frames
    narrower than the kernel, single rows / columns / frames, tiles that span several clips, ragged cout, asymmetric front pads,
    residual on and off. 27 must agree bit for bit; 28 walks K chunk-major: within one f16 rounding step."""
    from ted_spad_amd import _lib, engine as E
    rng = np.random.RandomState(1234)
    tdt = torch.float16
    cases = []
    for _ in range(10):
        kh, kw = int(rng.choice([1, 2, 3])), int(rng.choice([2, 3]))
        cases.append((27, (int(rng.randint(1, 5)), int(rng.randint(1, 4)), int(rng.randint(1, 12)), int(rng.randint(1, 20))), 64,
                      int(rng.choice([8, 24, 40, 64])), (1, kh, kw), (0, int(rng.randint(0, kh)), int(rng.randint(0, kw))), bool(rng.randint(2))))
    for _ in range(10):
        kt = int(rng.choice([2, 3]))
        cases.append((28, (int(rng.randint(1, 5)), int(rng.randint(1, 5)), int(rng.randint(1, 12)), int(rng.randint(1, 14))), int(rng.choice([64, 128, 256])),
                      int(rng.choice([8, 32, 56, 64])), (kt, 1, 1), (int(rng.randint(0, kt)), 0, 0), bool(rng.randint(2))))
    ran = {27: 0, 28: 0}
    for i, (cfg, dims, cin, cout, k, pf, use_res) in enumerate(cases):
        n, t, h, w = dims
        pb = tuple(k[d] - 1 - pf[d] for d in range(3))                 # 'same' output extent with an asymmetric split of the padding
        x = synth_tensor(21, "fz_x%d" % i, (n, t, h, w, cin), -1, 1).to(tdt)
        wgt = (synth_tensor(21, "fz_w%d" % i, (cout, cin) + k, -1, 1) * (2.0 / (cin * k[0] * k[1] * k[2])) ** 0.5).to(tdt).float()
        scale, shift = synth_tensor(21, "fz_s%d" % i, (cout,), 0.5, 1.5), synth_tensor(21, "fz_b%d" % i, (cout,), -0.3, 0.3)
        pc = E.PackedConv(wgt, scale, shift, dtype="f16", device="cuda")
        xa = E.Act(x.cuda(), cin)
        ra = E.Act(synth_tensor(21, "fz_r%d" % i, (n, t, h, w, pc.cout), -1, 1).to(tdt).cuda(), pc.cout) if use_res else None
        outs = {}
        try:
            for c in (cfg, 5):
                E.FORCE_TILE_CFG = c
                outs[c] = pc(xa, pads=pf, pads_back=pb, residual=ra, relu=True).buf.float().cpu()
        except _lib.TedSpadHipError as e:
            assert c == cfg, (c, str(e))        # the generic tile always applies; the flat tiles may decline (e.g. T > 4)
            continue
        finally:
            E.FORCE_TILE_CFG = None
        ran[cfg] += 1
        if cfg == 27:
            assert torch.equal(outs[27], outs[5]), (cfg, dims, cin, cout, k, pf)
        else:
            assert bool(((outs[28] - outs[5]).abs() <= 2.0 ** -10 * outs[5].abs() + 1e-4).all()), (cfg, dims, cin, cout, k, pf)
    assert ran[27] >= 6 and ran[28] >= 5, ran


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("dims,chans,cout,k", [((3, 1, 36, 40), (64, 64), 64, (1, 3, 3)), ((2, 1, 28, 28), (128, 64, 64), 128, (1, 3, 3)), ((1, 2, 18, 50), (64,), 32, (1, 3, 3)),
                                               ((2, 1, 64, 72), (128, 64, 64, 64), 64, (1, 3, 3)), ((5, 1, 14, 14), (256, 128), 256, (1, 3, 3)), ((1, 1, 2, 2), (64, 128), 24, (1, 3, 3))])
def test_conv_on_a_gathered_concatenation_equals_the_materialised_one(dims, chans, cout, k, dtype):
    """PackedConv.gather (tedspad_conv_extras.nchunk_src): the first conv of a unet++ decoder block reads `cat([interpolate(x, 2, 'nearest'), *skips])`
    in place -- source 0 at half resolution through the x2 index map, the skips from tensors with their own pixel strides (slices of wider
    buffers). Against the same tile configuration on the materialised concat buffer: bit for bit (same LDS image, same K order); against the
    oracle's conv on the torch-built concatenation: the conv tolerance."""
    from oracle.conv_ref import conv_cl
    from ted_spad_amd import _lib, engine as E
    tdt = E.DTYPES[dtype][0]
    n, t, h, w = dims
    cin = sum(chans)
    pf = (0, (k[1] - 1) // 2, (k[2] - 1) // 2)
    pb = (0, k[1] - 1 - pf[1], k[2] - 1 - pf[2])
    wgt = (synth_tensor(31, "gc_w", (cout, cin) + k, -1, 1) * (2.0 / (cin * k[1] * k[2])) ** 0.5).to(tdt).float()
    scale, shift = synth_tensor(31, "gc_s", (cout,), 0.5, 1.5), synth_tensor(31, "gc_b", (cout,), -0.3, 0.3)
    pc = E.PackedConv(wgt, scale, shift, dtype=dtype, device="cuda")
    srcs, parts = [], []
    for i, c in enumerate(chans):
        hh, ww = (h // 2, w // 2) if i == 0 else (h, w)
        wide = synth_tensor(31, "gc_x%d" % i, (n, t, hh, ww, c + 16 * i), -1, 1).to(tdt).cuda()        # the skips are channel slices of wider buffers
        a = E.Act(wide, c, 8 * i)
        srcs.append((a, i == 0))
        v = wide[..., 8 * i:8 * i + c]
        parts.append(v.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3) if i == 0 else v)
    cat = torch.cat(parts, dim=-1).contiguous()
    ref = conv_cl(cat.float().cpu(), wgt, scale, shift, (1, 1, 1), pf, pb, None, True)
    ran = 0
    for cfg in (32, 33, 38, 40):
        E.FORCE_TILE_CFG = cfg
        try:
            want = pc(E.Act(cat, cin), pads=pf, pads_back=pb).buf
            got = pc.gather(srcs, pads=pf).buf
        except _lib.TedSpadHipError:
            continue                                  # the flat form declines wide frames
        finally:
            E.FORCE_TILE_CFG = None
        ran += 1
        assert torch.equal(got, want), (cfg, dims, chans)
        assert rel_l2(got.float().cpu()[..., :cout], ref) < (2e-3 if dtype == "f16" else 1.2e-2), cfg
    assert ran >= 1
    got = pc.gather(srcs, pads=pf).buf                # default path (tuner / heuristic)
    assert rel_l2(got.float().cpu()[..., :cout], ref) < (2e-3 if dtype == "f16" else 1.2e-2)
    E.FORCE_TILE_CFG = 5
    try:
        with pytest.raises(_lib.TedSpadHipError):     # the generic tiles do not take gathered sources
            pc.gather(srcs, pads=pf)
    finally:
        E.FORCE_TILE_CFG = None


@pytest.mark.parametrize("dims,c1,c2,cout,stride", [((3, 2, 14, 14), 128, 256, 512, 2), ((2, 2, 7, 9), 256, 512, 1024, 2), ((1, 3, 5, 5), 64, 64, 256, 1),
                                                    ((2, 1, 28, 27), 128, 64, 256, 2)])
def test_dual_p8_k_concatenated_pair(dims, c1, c2, cout, stride):
    """tedspad_conv_p8_dual_fwd (conv3 + bn3 and the STRIDED downsample branch of layer2.0 / 3.0 / 4.0 as one GEMM over
    [W3*s3 | Wd*sd]) against the oracle's two convolutions summed in fp32 and against the two launches it replaces."""
    from oracle.conv_ref import conv_cl
    from ted_spad_amd import engine as E
    n, t, h, w = dims
    h2, w2 = (h - 1) * stride + 1 + (stride - 1), (w - 1) * stride + 1      # odd / even source grids
    tdt = torch.float16
    x = synth_tensor(12, "px", (n, t, h, w, c1), -1, 1).to(tdt).float()
    x2 = synth_tensor(12, "px2", (n, t, h2, w2, c2), -1, 1).to(tdt).float()
    w1 = (synth_tensor(12, "pw1", (cout, c1, 1, 1, 1), -1, 1) * (2.0 / c1) ** 0.5).to(tdt).float()
    w2_ = (synth_tensor(12, "pw2", (cout, c2, 1, 1, 1), -1, 1) * (2.0 / c2) ** 0.5).to(tdt).float()
    s1, b1 = synth_tensor(12, "ps1", (cout,), 0.5, 1.5), synth_tensor(12, "pb1", (cout,), -0.3, 0.3)
    s2, b2 = synth_tensor(12, "ps2", (cout,), 0.5, 1.5), synth_tensor(12, "pb2", (cout,), -0.3, 0.3)
    z = (0, 0, 0)
    a = conv_cl(x, w1, s1, b1, (1, 1, 1), z, z, None, relu=False)
    b = conv_cl(x2, w2_, s2, b2, (1, stride, stride), z, z, None, relu=False)[:, :, :h, :w]
    ref = torch.relu(a + b)
    pc = E.PackedConv.fused_pair(w1, s1, b1, w2_, s2, b2, dtype="f16", device="cuda")
    xa, x2a = E.Act(x.to(tdt).cuda(), c1), E.Act(x2.to(tdt).cuda(), c2)
    assert pc.dual_p8_supported(xa, x2a, (stride, stride))
    got = pc.call_dual_p8(xa, x2a, (stride, stride), relu=True).buf.float().cpu()
    torch.cuda.synchronize()
    assert got.shape == ref.shape
    # the BatchNorm scales are folded into the 16-bit weights here (one more rounding per weight than the two-launch path)
    err = (got - ref).abs()
    assert bool((err <= 2.0 ** -9 * (a.abs() + b.abs()) + 2e-3).all()), "max err %g" % float(err.max())
    assert rel_l2(got, ref) < 5e-4


@pytest.mark.parametrize("dims,cin,cout,use_res", [((3, 4, 11, 13), 64, 256, True), ((2, 5, 7, 9), 128, 72, False),
                                                   ((2, 2, 16, 16), 64, 64, True), ((1, 8, 30, 30), 64, 256, True)])
def test_conv_pool_t2_fused_equals_conv_then_pool(dims, cin, cout, use_res):
    """tedspad_conv_pool_t2_fwd (1x1x1 conv + BN + residual + ReLU + MaxPool3d((2,1,1),(2,1,1)) in one persistent launch)
    must equal the two separate launches BIT-EXACTLY: rounding to f16 is monotonic, so max-then-round == round-then-max.
    Covers frames whose pixel count is not a multiple of the 128-pixel tile, an odd frame count (last frame dropped,
    as nn.MaxPool3d does) and a ragged channel tile."""
    from ted_spad_amd import engine as E
    n, t, h, w = dims
    x = synth_tensor(11, "ptx", (n, t, h, w, cin), -1, 1).half()
    wgt = (synth_tensor(11, "ptw", (cout, cin, 1, 1, 1), -1, 1) * (2.0 / cin) ** 0.5)
    scale, shift = synth_tensor(11, "pts", (cout,), 0.5, 1.5), synth_tensor(11, "ptb", (cout,), -0.3, 0.3)
    res = synth_tensor(11, "ptr", (n, t, h, w, cout), -1, 1).half() if use_res else None
    pc = E.PackedConv(wgt, scale, shift, dtype="f16", device="cuda")
    xa, ra = E.Act(x.cuda(), cin), (E.Act(res.cuda(), cout) if use_res else None)
    two = E.maxpool(pc(xa, residual=ra, relu=True), (2, 1, 1), (2, 1, 1))
    one = pc.call_pool_t2(xa, residual=ra, relu=True)
    assert one.dims == two.dims == (n, t // 2, h, w)
    assert torch.equal(one.buf, two.buf)


@pytest.mark.parametrize("dims", [(3, 16, 224, 224), (2, 6, 100, 76), (1, 2, 32, 64), (5, 10, 48, 40)])
def test_two_frame_stem_equals_halo_stem(dims):
    """tile_cfg 20 (halo-direct stem on two output frames per workgroup, 8 waves sharing each weight stage) against
    tile_cfg 9 and a generic configuration on the 5x7x7 / stride-2 stem: bit-exact, including an odd number of output
    frames (the second half of the last patch is masked) and patches that hang over the right / bottom edge."""
    from ted_spad_amd import engine as E
    n, t, h, w = dims
    wgt = synth_tensor(13, "stw", (64, 3, 5, 7, 7), -1, 1) * (2.0 / 735) ** 0.5
    scale, shift = synth_tensor(13, "sts", (64,), 0.5, 1.5), synth_tensor(13, "stb", (64,), -0.3, 0.3)
    pc = E.PackedConv(wgt, scale, shift, stride=(2, 2, 2), dtype="f16", device="cuda", pair_w=3)
    a = E.clip_to_act(synth_tensor(13, "stx", (n, 3, t, h, w), device="cuda"), cpad=4)
    outs = {}
    try:
        for cfg in (9, 20, 21, 29, 30, 31, 2):
            E.FORCE_TILE_CFG = cfg
            outs[cfg] = pc(a, pads=(2, 3, pc.pair_pw), pads_back=(2, 3, 1)).buf.clone()
    finally:
        E.FORCE_TILE_CFG = None
    assert torch.equal(outs[9], outs[2])
    assert torch.equal(outs[20], outs[9])
    assert torch.equal(outs[29], outs[9]) and torch.equal(outs[30], outs[21]) and torch.equal(outs[31], outs[9])    # 16 x 16 patches: same sums, other pixel -> lane map
    # 21: split-K over 8 waves -- fp32 partial sums re-associated: within one f16 rounding step
    o21, o9 = outs[21].float(), outs[9].float()
    assert bool(((o21 - o9).abs() <= 2.0 ** -10 * o9.abs() + 1e-4).all())
    assert float((o21 != o9).float().mean()) < 0.02


def test_hand_counted_waits_are_race_free_over_many_launches():
    """The persistent pointwise kernel (19, incl. the fused temporal pool), the chunk-major flat tiles (33) and the split-K
    stem (21) order their LDS reads behind LDS-DMA with hand-counted `s_waitcnt vmcnt(N)`; a wrong count shows up as a
    rare stale tile that comes and goes with timing. 60 launches each on large shapes, interleaved with a kernel that
    thrashes L2, must reproduce the first result bit for bit."""
    from ted_spad_amd import engine as E
    junk = torch.empty(64 << 20, dtype=torch.float16, device="cuda")

    def repeat(fn, n=60):
        first = fn().clone()
        for i in range(n):
            if i % 3 == 0:
                junk.normal_()                        # evict L2 / MALL between launches: different DMA latencies
            assert torch.equal(fn(), first), "launch %d differs" % i

    x = E.Act(synth_tensor(17, "rx", (24, 4, 55, 55, 64), -1, 1, device="cuda").half(), 64)
    r = E.Act(synth_tensor(17, "rr", (24, 4, 55, 55, 256), -1, 1, device="cuda").half(), 256)
    pc = E.PackedConv(synth_tensor(17, "rw", (256, 64, 1, 1, 1), -0.2, 0.2), synth_tensor(17, "rs", (256,), 0.5, 1.5),
                      synth_tensor(17, "rb", (256,), -0.3, 0.3), dtype="f16", device="cuda")
    x3 = E.Act(synth_tensor(17, "r3", (24, 2, 14, 14, 256), -1, 1, device="cuda").half(), 256)
    pc3 = E.PackedConv(synth_tensor(17, "rw3", (256, 256, 1, 3, 3), -0.05, 0.05), synth_tensor(17, "rs3", (256,), 0.5, 1.5),
                       synth_tensor(17, "rb3", (256,), -0.3, 0.3), dtype="f16", device="cuda")
    st = E.PackedConv(synth_tensor(17, "rws", (64, 3, 5, 7, 7), -0.1, 0.1), torch.ones(64), torch.zeros(64), stride=(2, 2, 2),
                      dtype="f16", device="cuda", pair_w=3)
    clip = E.clip_to_act(synth_tensor(17, "rc", (6, 3, 16, 224, 224), device="cuda"), cpad=4)
    try:
        E.FORCE_TILE_CFG = 19
        repeat(lambda: pc(x, residual=r, relu=True).buf)
        repeat(lambda: pc.call_pool_t2(x, residual=r, relu=True).buf)
        E.FORCE_TILE_CFG = 33
        repeat(lambda: pc3(x3, pads=(0, 1, 1)).buf)
        E.FORCE_TILE_CFG = 21
        repeat(lambda: st(clip, pads=(2, 3, st.pair_pw), pads_back=(2, 3, 1)).buf, n=30)
    finally:
        E.FORCE_TILE_CFG = None
    # the persistent stem (csrc/conv_stem_pt.hip): halo regions re-filled a phase ahead, the epilogue's stores left in flight
    # across the barrier by a counted wait; more patches than workgroups so that every workgroup walks several
    spt = E.StemPT(synth_tensor(17, "rws", (64, 3, 5, 7, 7), -0.1, 0.1), torch.ones(64), torch.zeros(64), dtype="f16", device="cuda")
    xtp = spt.layout(synth_tensor(17, "rc", (6, 3, 16, 224, 224), device="cuda"))
    for variant in (0, 2):
        repeat(lambda: spt.conv(xtp, variant=variant).buf, n=30)
    for variant in (2, 6):                               # ... with the whole pool fused, on 32x32x16 and on 16x16x32 MFMAs
        repeat(lambda: spt.conv_pool(xtp, variant=variant).buf, n=30)


def test_a_tuning_job_that_stopped_coming_no_longer_holds_the_streams(monkeypatch):
    """`engine.tuning_pending()` keeps multi-stream callers (bench.py's forwards, the weight-gradient side stream) on one stream while a tile
    is being timed. A geometry that is seen once and never again (a one-off batch size) would hold them there for the rest of the run: its job
    stops counting once TUNE_STALE further conv launches went by without it."""
    from ted_spad_amd import engine as E
    monkeypatch.setattr(E, "_TUNING", {})
    monkeypatch.setattr(E, "AUTOTUNE", True)
    wgt = synth_tensor(41, "tw", (64, 64, 1, 3, 3), -0.1, 0.1)
    pc = E.PackedConv(wgt, torch.ones(64), torch.zeros(64), dtype="f16", device="cuda")
    x = E.Act(synth_tensor(41, "tx", (1, 1, 9, 13, 64), -1, 1).to(torch.float16).cuda(), 64)
    pc(x, pads=(0, 1, 1))
    assert E.tuning_pending()
    monkeypatch.setattr(E, "_CLOCK", [E._CLOCK[0] + E.TUNE_STALE + 1])
    assert not E.tuning_pending()
    pc(x, pads=(0, 1, 1))                                     # it comes again: counted again
    assert E.tuning_pending()


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("temporal,n,t", [(False, 3, 2), (True, 3, 2), (False, 1, 5), (True, 1, 2)])
def test_whole_bottleneck_per_frame_vs_oracle_and_unfused(temporal, n, t, dtype, monkeypatch):
    """engine.BneckFrame (csrc/conv_bneck_frame.hip): conv1 (1x1x1 | 3x1x1) + bn1 + ReLU -> conv2 1x3x3 + bn2 + ReLU -> conv3 + bn3 + residual + ReLU of a plain
    layer3 bottleneck (large_i3d.py:61-84 without `downsample`; 14 x 14 frames, 1024 -> 256 -> 1024) in ONE launch, one workgroup per frame --
    against the oracle's three convolutions with both 256-channel tensors rounded where the unfused path stores them, and against the three
    launches it replaces (conv1 in its folded two-frame form for the temporal block)."""
    from oracle.conv_ref import conv_cl
    from ted_spad_amd import engine as E
    tdt = E.DTYPES[dtype][0]
    name = "bf%d%d%d" % (temporal, n, t)
    kt = 3 if temporal else 1
    x = _round(synth_tensor(11, name + "x", (n, t, 14, 14, 1024), -1, 1), tdt)
    w1 = _round(synth_tensor(11, name + "w1", (256, 1024, kt, 1, 1), -1, 1) * (2.0 / (1024 * (2 if temporal else 1))) ** 0.5, tdt)
    w2 = _round(synth_tensor(11, name + "w2", (256, 256, 1, 3, 3), -1, 1) * (2.0 / 2304) ** 0.5, tdt)
    w3 = _round(synth_tensor(11, name + "w3", (1024, 256, 1, 1, 1), -1, 1) * (2.0 / 256) ** 0.5, tdt)
    bn = [synth_tensor(11, name + "bn%d" % i, (c,), lo, hi) for i, (c, lo, hi) in enumerate(
        [(256, 0.5, 1.5), (256, -0.3, 0.3), (256, 0.5, 1.5), (256, -0.3, 0.3), (1024, 0.5, 1.5), (1024, -0.3, 0.3)])]
    pt = 1 if temporal else 0
    m1 = _round(conv_cl(x, w1, bn[0], bn[1], (1, 1, 1), (pt, 0, 0), (pt, 0, 0), None, relu=True), tdt)
    m2 = _round(conv_cl(m1, w2, bn[2], bn[3], (1, 1, 1), (0, 1, 1), (0, 1, 1), None, relu=True), tdt)
    ref = conv_cl(m2, w3, bn[4], bn[5], (1, 1, 1), (0, 0, 0), (0, 0, 0), x, relu=True)
    bf = E.BneckFrame(w1, bn[0], bn[1], w2, bn[2], bn[3], w3, bn[4], bn[5], dtype=dtype, device="cuda")
    xa = E.Act(x.to(tdt).cuda(), 1024)
    assert bf.applies(xa) and bf.temporal == temporal
    got = bf(xa).buf.float().cpu()
    assert got.shape == ref.shape
    ulp = 2.0 ** -10 if dtype == "f16" else 2.0 ** -7
    err = (got - ref).abs()
    # a 16-bit intermediate that lands on the other side of a rounding boundary moves the output by more than its own rounding step
    assert rel_l2(got, ref) < (5e-4 if dtype == "f16" else 4e-3), rel_l2(got, ref)
    assert float(err.max()) < (0.02 if dtype == "f16" else 0.15), "max err %g" % float(err.max())
    assert float((err <= ulp * ref.abs() + 2e-3).float().mean()) > (0.97 if dtype == "f16" else 0.5)
    # ... and the three launches it replaces
    monkeypatch.setattr(E, "FORCE_TILE_CFG", None)
    cuda = lambda v: v.cuda()
    if temporal:
        h1 = E.TPairConv(w1, bn[0], bn[1], dtype=dtype, device="cuda")(xa)
    else:
        h1 = E.PackedConv(w1, cuda(bn[0]), cuda(bn[1]), dtype=dtype, device="cuda")(xa, pads=(0, 0, 0))
    h2 = E.PackedConv(w2, cuda(bn[2]), cuda(bn[3]), dtype=dtype, device="cuda")(h1, pads=(0, 1, 1))
    old = E.PackedConv(w3, cuda(bn[4]), cuda(bn[5]), dtype=dtype, device="cuda")(h2, residual=xa, relu=True).buf.float().cpu()
    assert rel_l2(got, old) < (6e-4 if dtype == "f16" else 5e-3), rel_l2(got, old)
    # a second launch gives the same bits (no race between the DMA rings and the fragment reads)
    assert torch.equal(bf(xa).buf, bf(xa).buf)


def test_whole_bottleneck_refuses_other_geometries():
    from ted_spad_amd import engine as E
    w1, w2, w3 = torch.zeros(256, 1024, 1, 1, 1), torch.zeros(256, 256, 1, 3, 3), torch.zeros(1024, 256, 1, 1, 1)
    assert E.BneckFrame.supported(w1, w2, w3) and not E.BneckFrame.supported(torch.zeros(128, 512, 1, 1, 1), w2, w3)
    one = torch.ones(256)
    bf = E.BneckFrame(w1, one, one, w2, one, one, w3, torch.ones(1024), torch.ones(1024))
    assert not bf.applies(E.Act(torch.zeros(1, 2, 7, 7, 1024, dtype=torch.float16, device="cuda"), 1024))
    assert not bf.applies(E.Act(torch.zeros(1, 2, 14, 14, 512, dtype=torch.float16, device="cuda"), 512))
