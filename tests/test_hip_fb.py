"""-m gpu: the privacy branch fb (ResNet-50 + MLP, aux_code/model_loaders.py:124-153) on MI355X against the CPU
oracle oracle/resnet50_ref.py (torchvision's ResNet-50 restated -- that trunk's parity is unpinned, see its header):
eval forward, both backward flavours, and the fb terms of the two training phases."""
import numpy as np
import pytest
import torch

from conftest import rel_l2
from ted_spad_amd.synth import synth_state_dict, synth_tensor
from test_hip_train_step import _grad_sd, _models, _report, _smooth

pytestmark = pytest.mark.gpu


def _fb(beta=None):
    from ted_spad_amd.model_loaders import load_fb_model
    fb = load_fb_model(arch="r50", ssl=True)
    sd = synth_state_dict(fb.state_dict(), 0)
    if beta is not None:
        sd = _smooth(sd, beta)
    fb.load_state_dict(sd)
    return fb.cuda(), sd


def test_fb_eval_forward_vs_oracle():
    """Tolerance: 1e-3 rel-L2 on the 2048-d pooled feature (the same gate as the I3D clip feature), 2e-3 on the
    128-d unit-norm embedding (two more fp32 GEMVs + normalisation)."""
    from oracle import resnet50_ref
    fb, sd = _fb()
    fb.eval()
    x = synth_tensor(0, "vispr", (4, 3, 224, 224))
    f_ref, z_ref = resnet50_ref.trunk(x, sd), resnet50_ref.forward(x, sd)
    f, z = fb[0](x.cuda()), fb(x.cuda())
    assert f.shape == (4, 2048) and z.shape == (4, 128)
    assert rel_l2(f.cpu(), f_ref) < 1e-3
    assert rel_l2(z.cpu(), z_ref) < 2e-3
    assert torch.allclose(z.norm(dim=1).cpu(), torch.ones(4), atol=1e-5)


def test_fb_predictor_head_and_errors():
    from ted_spad_amd.model_loaders import load_fb_model
    from ted_spad_amd._lib import TedSpadHipError
    fb = load_fb_model(arch="r50", ssl=False, num_pa=7)
    fb.load_state_dict(synth_state_dict(fb.state_dict(), 0))
    fb = fb.cuda().eval()
    assert fb(synth_tensor(0, "pa", (2, 3, 64, 64)).cuda()).shape == (2, 7)
    with pytest.raises(TedSpadHipError):
        fb(torch.zeros(2, 3, 64, 64))                       # no CPU path
    assert load_fb_model(arch="r18") is None                # model_loaders.py:101-103: print + None


def test_fb_backward_chains_tight_on_a_smooth_network():
    """FBTrainer train-mode chain (parameter gradients) and eval-mode chain (gradient w.r.t. the image), BN bias +4
    so that ReLU flips do not mask a wrong kernel (see test_hip_train_step.py)."""
    from oracle import resnet50_ref
    from ted_spad_amd.train_nets import FBTrainer
    fb, sd = _fb(beta=4.0)
    x = synth_tensor(0, "fbx", (6, 3, 64, 64)) * (torch.arange(1, 7).float() / 6).view(6, 1, 1, 1)
    dz = synth_tensor(0, "fbdz", (6, 128), -1, 1)
    tr = FBTrainer(fb)
    sdg = _grad_sd(sd)
    z_ref = resnet50_ref.forward(x, sdg, train=True)
    (z_ref * dz).sum().backward()
    fb.train()
    z, tape = tr.forward(x.cuda(), "train")
    assert rel_l2(z.cpu(), z_ref.detach()) < 5e-3
    tr.backward(tape, dz.cuda())
    tr.flush_grads()
    errs = _report("fb train chain (smooth)", {k: q.grad for k, q in fb.named_parameters()}, {k: v.grad for k, v in sdg.items() if v.requires_grad},
                   min_cos=0.95, med_cos=0.998, tiny=5e-3)
    # 6 images of 64x64: layer4 normalises over 6 x 2 x 2 = 24 values per channel, which amplifies the 16-bit storage error
    # (run-to-run spread of this median with the float-atomic order of the batch statistics: 0.03 .. 0.045)
    assert float(np.median(list(errs.values()))) < 7e-2
    assert int(fb[0].bn1.num_batches_tracked) == 1
    # eval mode: d(image)
    fb.load_state_dict(sd)
    fb.eval()
    xg = x.clone().requires_grad_()
    (resnet50_ref.forward(xg, sd) * dz).sum().backward()
    z, tape = tr.forward(x.cuda(), "eval")
    dx = tr.backward(tape, dz.cuda())
    e = rel_l2(dx.cpu(), xg.grad)
    print("fb eval chain: d(image) rel-L2 %.3e" % e)
    assert e < 0.25                                           # ReLU-flip error of a 49-ReLU eval chain, as for I3Res50


def test_train_step_with_fb_both_phases():
    """The whole train_epoch body: phase 1 with the -NT-Xent privacy term through the frozen fb, phase 2 updating fb."""
    from oracle import train_step_ref
    from ted_spad_amd.train_step import AnonymizerTrainStep
    from ted_spad_amd.synth import synth_train_video
    fa, ft, sd_u, sd_l = _models()
    fb, sd_b = _fb()
    video = synth_train_video(0, "train_video", (2, 48, 3, 32, 32))
    labels = torch.tensor([5, 77])
    gain = (torch.arange(1, 5).float() / 4).view(4, 1, 1, 1)
    vispr = [synth_tensor(0, "vispr%d" % i, (4, 3, 128, 128)) * gain for i in range(2)]   # layer4 of fb: 4x4x4 values per channel for its train-mode BN
    sd_u1 = {k: v.clone() for k, v in sd_u.items()}
    ref_l, ref_g, _ = train_step_ref.phase1(video, labels, sd_u1, sd_l, vispr=vispr, fb_sd=sd_b)
    step = AnonymizerTrainStep(fa, ft, fb_model=fb)
    fb_before = {k: v.detach().clone() for k, v in fb.state_dict().items()}
    out = step.step_fa(video.cuda(), labels.cuda(), [v.cuda() for v in vispr])
    assert abs(out["loss_fb"] - ref_l["loss_fb"]) < 5e-3 * abs(ref_l["loss_fb"])
    assert abs(out["loss_fa"] - ref_l["loss_fa"]) < 5e-3 * max(abs(ref_l["loss_fa"]), abs(ref_l["loss_fb"]))
    errs = _report("phase1+fb fa grads", {k: p.grad for k, p in fa.named_parameters()}, ref_g)
    assert float(np.median(list(errs.values()))) < 0.3 and max(errs.values()) < 0.5
    assert all(torch.equal(v, fb_before[k]) for k, v in fb.state_dict().items())       # fb frozen in phase 1
    assert int(fa.inc.double_conv[1].num_batches_tracked) == 3                          # fa saw view0, view1, the video (Q14)
    with pytest.raises(ValueError):
        step.step_fa(video.cuda(), labels.cuda())                                       # fb given but no views
    # ---- phase 2: fb is updated with the NT-Xent loss of the (eval-mode) anonymised views
    fa2, ft2, sd_u, sd_l = _models()
    fb2, sd_b = _fb()
    ref_loss, ref_gb = train_step_ref.phase2_fb(vispr, sd_u, {k: v.clone() for k, v in sd_b.items()})
    step2 = AnonymizerTrainStep(fa2, ft2, fb_model=fb2)
    video4 = synth_train_video(0, "train_video64", (4, 48, 3, 64, 64))
    out = step2.step_ft(video4.cuda(), torch.tensor([5, 77, 101, 1]).cuda(), inputs_vispr=[v.cuda() for v in vispr])
    assert out["phase"] == 2 and abs(out["loss_fb"] - ref_loss) < 1e-2 * abs(ref_loss)
    # Conditioning: a randomly initialised fb maps all views to nearly parallel embeddings (pairwise cosine 0.90-0.96),
    # and the NT-Xent gradient sum_j (p_ij - t_ij) z_j cancels their common component exactly, so the 2e-3 forward error
    # of z is already a few % of d(loss)/dz (cosine 0.94-0.99 at fc1/fc2); the batch-of-4 train-mode BNs below amplify
    # it like in the phase-2 ft test. The kernels themselves are held to the tight bounds of the smooth-chain test
    # above; here the direction of every parameter gradient and the loss value are checked.
    errs = _report("phase2 fb grads", {k: p.grad for k, p in fb2.named_parameters()}, ref_gb, min_cos=0.5, med_cos=0.75, tiny=1e-3)
    assert float(np.median(list(errs.values()))) < 0.7
    moved = max(float((p.detach().cpu() - sd_b[k]).abs().max()) for k, p in fb2.named_parameters())
    assert 0 < moved <= 1.05 * step2.params.learning_rate_fb                             # Adam's first step: lr * sign(grad)
    assert int(fb2[0].bn1.num_batches_tracked) == 2                                       # one train-mode forward per view


def test_the_two_views_as_one_batch_equal_two_passes(deterministic):
    """train_anonymizer.py:80-84,153-157 call fa / fb once per VISPR view; AnonymizerTrainStep runs both views as ONE batch whose BatchNorms keep a set of
    batch statistics per view (groups = 2: tedspad_conv_extras.stats_rows, the BatchNorm kernels' `groups`) and move the running statistics view by view.
    Same weights, same inputs, batched vs two passes, in deterministic mode (one K-order-preserving tile per conv: every conv output is the same dot
    product either way; only the order in which the tiles' partial sums reach the batch statistics differs): losses, every parameter gradient of the
    network being updated, the running statistics and their step counters."""
    from ted_spad_amd.train_step import AnonymizerTrainStep
    from ted_spad_amd.synth import synth_train_video
    video = synth_train_video(0, "train_video", (2, 48, 3, 32, 32)).cuda()
    labels = torch.tensor([5, 77]).cuda()
    gain = (torch.arange(1, 17).float() / 16).view(16, 1, 1, 1)
    vispr = [(synth_tensor(0, "vispr_b%d" % i, (16, 3, 128, 128)) * gain).cuda() for i in range(2)]   # 16 x 4 x 4 = 256 values per channel and view in fb's layer4
    runs = {}
    for batched in (True, False):
        fa, ft, _, _ = _models()
        fb, _ = _fb()
        step = AnonymizerTrainStep(fa, ft, fb_model=fb)
        assert step._views_batchable(vispr)
        step.batch_views = batched
        o1 = step.step_fa(video, labels, vispr)
        g_fa = {k: p.grad.detach().clone() for k, p in fa.named_parameters() if p.grad is not None}
        s_fa = {k: v.detach().clone() for k, v in fa.state_dict().items()}
        fa, ft, _, _ = _models()              # phase 2 from the same fresh weights too (Adam's first step is lr * sign(grad): it would amplify phase 1's last bits)
        fb, _ = _fb()
        step = AnonymizerTrainStep(fa, ft, fb_model=fb)
        step.batch_views = batched
        o2 = step.step_ft(video, labels, inputs_vispr=vispr)
        g_fb = {k: p.grad.detach().clone() for k, p in fb.named_parameters() if p.grad is not None}
        runs[batched] = (o1, g_fa, o2, g_fb, s_fa, {k: v.detach().clone() for k, v in fb.state_dict().items()})
    (a1, ga, a2, gb, sa, sb), (b1, ha, b2, hb, ta, tb) = runs[True], runs[False]
    for k in ("loss_fa", "loss_fb"):
        assert abs(a1[k] - b1[k]) < 1e-4 * max(1.0, abs(b1[k])), (k, a1[k], b1[k])
    assert abs(a2["loss_fb"] - b2["loss_fb"]) < 1e-4 * abs(b2["loss_fb"])
    assert set(ga) == set(ha) and set(gb) == set(hb) and len(gb) > 150
    ea = {k: rel_l2(ga[k].float().cpu(), ha[k].float().cpu()) for k in ga}
    eb = {k: rel_l2(gb[k].float().cpu(), hb[k].float().cpu()) for k in gb}
    print("fa worst", sorted(ea.items(), key=lambda kv: -kv[1])[:4], "median", float(np.median(list(ea.values()))))
    print("fb worst", sorted(eb.items(), key=lambda kv: -kv[1])[:4], "median", float(np.median(list(eb.values()))))
    # fa (phase 1): every tensor but the conv biases in front of a BatchNorm (their gradient is zero in exact arithmetic: what is left is rounding noise)
    assert float(np.median(list(ea.values()))) < 1e-3 and max(v for k, v in ea.items() if not k.endswith("double_conv.0.bias") and not k.endswith("double_conv.3.bias")) < 1e-2
    # fb (phase 2): the NT-Xent gradient of a randomly initialised fb is ill-conditioned (test_train_step_with_fb_both_phases): the last bits of the batch
    # statistics already move it; the grouped chain itself is held tight on the smooth network below. Here: norms and directions.
    for k in gb:
        a, b = gb[k].float().flatten().cpu(), hb[k].float().flatten().cpu()
        assert abs(float(a.norm()) - float(b.norm())) < 0.15 * float(b.norm()) + 1e-6, k
    assert float(np.median(list(eb.values()))) < 0.5
    for s, t in ((sa, ta), (sb, tb)):
        for k in s:
            if k.endswith("num_batches_tracked"):
                assert int(s[k]) == int(t[k]), k
            elif "running_" in k:
                assert rel_l2(s[k].float().cpu(), t[k].float().cpu()) < 1e-3, k
    assert int(sa["inc.double_conv.1.num_batches_tracked"]) == 3 and int(sb["0.bn1.num_batches_tracked"]) == 2


def test_fb_grouped_train_chain_equals_separate_calls_on_a_smooth_network(deterministic):
    """FBTrainer.forward(x, 'train', groups=2) + backward against two separate calls on the two halves (gradients accumulated), BN bias +4 (no ReLU flips):
    embeddings, parameter gradients, the running statistics after both blocks. The two ways are NOT bit-equal even in deterministic mode: a launch over 32 images
    tiles and sums its one-pass batch statistics (sum, sum of squares -> E[z^2] - mean^2) in another order than two launches over 16, and the cancellation in the
    variance turns that into a 1e-4 change of invstd, i.e. 16-bit roundings that fall the other way all along the chain. Measured: embeddings 2.8e-4 apart on every
    row (two separate runs: 0), weight gradients 1.3e-2 .. 3e-2 (the spread this chain shows run to run against the oracle, see the smooth-chain test above);
    the BatchNorm biases in front of another BatchNorm have a zero gradient in exact arithmetic and are left out."""
    from ted_spad_amd.train_nets import FBTrainer
    x = (synth_tensor(0, "fbgx", (32, 3, 128, 128)) * (torch.arange(1, 33).float() / 32).view(32, 1, 1, 1)).cuda()
    dz = synth_tensor(0, "fbgdz", (32, 128), -1, 1).cuda()
    res = {}
    for grouped in (True, False):
        fb, _ = _fb(beta=4.0)
        fb.train()
        tr = FBTrainer(fb)
        if grouped:
            z, tape = tr.forward(x, "train", groups=2)
            tr.backward(tape, dz)
        else:
            zs = []
            tapes = [tr.forward(x[h * 16:(h + 1) * 16], "train") for h in range(2)]
            for h, (zh, tape) in enumerate(tapes):
                zs.append(zh)
                tr.backward(tape, dz[h * 16:(h + 1) * 16])
            z = torch.cat(zs)
        tr.flush_grads()
        res[grouped] = (z.detach().cpu(), {k: q.grad.detach().float().cpu() for k, q in fb.named_parameters()}, {k: v.detach().float().cpu() for k, v in fb.state_dict().items()})
    (za, ga, sa), (zb, gb, sb) = res[True], res[False]
    assert rel_l2(za, zb) < 1e-3
    errs = {k: rel_l2(ga[k], gb[k]) for k in ga}
    print("fb grouped vs separate (smooth): worst", sorted(errs.items(), key=lambda kv: -kv[1])[:4], "median", float(np.median(list(errs.values()))))
    real = {k: v for k, v in errs.items() if not (k.endswith(".bias") and (".bn" in k or ".downsample.1." in k))}
    assert float(np.median(list(real.values()))) < 5e-2 and max(real.values()) < 0.15, sorted(real.items(), key=lambda kv: -kv[1])[:4]
    for k in sa:
        if k.endswith("num_batches_tracked"):
            assert int(sa[k]) == int(sb[k]) == 2, k
        elif "running_" in k:
            assert rel_l2(sa[k], sb[k]) < 1e-3, k
