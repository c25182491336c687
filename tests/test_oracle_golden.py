"""Pins the oracle (oracle/*.py, the CPU restatement) against golden vectors captured by
RUNNING THE REFERENCE (tests/golden/make_golden.py). CPU only."""
import io
import os

import numpy as np
import pytest
import torch

from conftest import rel_l2
from oracle import extract_ref, i3res50_ref, inception_i3d_ref, losses_ref, unet_ref
import reference_shapes as RS
from ted_spad_amd.synth import synth_clips, synth_state_dict, synth_tensor, synth_train_video

SEED = 0
TOL = 2e-5  # fp32 CPU restatement vs fp32 reference (different op order only)


@pytest.fixture(scope="module")
def sd_largei3d():
    return synth_state_dict(RS.wrapper_i3d_template(102), SEED)


@pytest.fixture(scope="module")
def sd_inception():
    return synth_state_dict(RS.inception_i3d_template(102), SEED)


@pytest.fixture(scope="module")
def sd_unet():
    return synth_state_dict(RS.unet_template(), SEED)


def _i3d(sd):
    return {k[4:]: v for k, v in sd.items() if k.startswith("i3d.")}


def test_state_dict_inventory(golden_meta, sd_largei3d, sd_inception, sd_unet):
    # entry counts / parameter counts of the reference modules
    assert len(sd_largei3d) == golden_meta["wrapper_state_dict_keys"] == 333
    assert len(sd_inception) == golden_meta["inception_state_dict_keys"] == 344
    assert len(sd_unet) == golden_meta["unet_state_dict_keys"] == 128
    nparam = lambda sd: sum(v.numel() for k, v in sd.items()
                            if not k.endswith(("running_mean", "running_var", "num_batches_tracked")))
    assert nparam(sd_largei3d) == golden_meta["wrapper_params"] == 28548774
    assert nparam(sd_inception) == golden_meta["inception_params"] == 12391814
    assert nparam(sd_unet) == golden_meta["unet_params"] == 17267523


def test_i3res50_cfg1_112(golden, sd_largei3d):
    x = synth_clips(SEED, 1, (3, 16, 112, 112))
    with torch.no_grad():
        f = i3res50_ref.extract_features(x, _i3d(sd_largei3d))
    assert f.shape == (1, 2048, 1, 1, 1)
    assert rel_l2(f.reshape(1, 2048), golden["i3res50_feat_112"]) < TOL


def test_i3res50_224_and_taps(golden, golden_meta, sd_largei3d):
    x = synth_clips(SEED, 2, (3, 16, 224, 224))
    taps = {}
    with torch.no_grad():
        f = i3res50_ref.extract_features(x, _i3d(sd_largei3d), taps=taps)
    for i in range(2):
        assert rel_l2(f[i].reshape(-1), golden["i3res50_feat_224"][i]) < TOL
    for name, (mean, l2) in golden_meta["i3res50_taps_224"].items():
        t = taps[name].double()
        assert abs(float(t.mean()) - mean) <= 1e-4 * abs(mean) + 1e-6, name
        assert abs(float(t.norm()) - l2) <= 1e-4 * l2, name


def test_wrapper_forward_eval_and_train(golden, sd_largei3d):
    x = synth_clips(SEED, 2, (3, 16, 112, 112))
    with torch.no_grad():
        pred, feat = i3res50_ref.wrapper_forward(x, sd_largei3d, train=False)
        assert rel_l2(pred, golden["wrapper_eval_pred"]) < TOL
        assert rel_l2(feat, golden["wrapper_eval_feat"]) < 1e-4
        pred, feat = i3res50_ref.wrapper_forward(x, sd_largei3d, train=True)
        assert rel_l2(pred, golden["wrapper_train_pred"]) < 1e-3  # batch-stat BN on B=2
        assert rel_l2(feat, golden["wrapper_train_feat"]) < 1e-3


def test_inception_224_and_taps(golden, golden_meta, sd_inception):
    x = synth_clips(SEED, 2, (3, 16, 224, 224))
    taps = {}
    with torch.no_grad():
        f = inception_i3d_ref.extract_features(x, sd_inception, taps=taps)
    assert f.shape == (2, 1024, 1, 1, 1)
    for i in range(2):
        assert rel_l2(f[i].reshape(-1), golden["inception_feat_224"][i]) < TOL
    for name, (mean, l2) in golden_meta["inception_taps_224"].items():
        t = taps[name].double()
        assert abs(float(t.norm()) - l2) <= 1e-4 * l2, name


def test_inception_forward_112_and_q4(golden, golden_meta, sd_inception):
    x = synth_clips(SEED, 2, (3, 16, 112, 112))
    with torch.no_grad():
        lg = inception_i3d_ref.forward(x, sd_inception)
    assert lg.shape == (2, 102)  # ONE tensor, not a tuple (Q6)
    assert rel_l2(lg, golden["inception_logits_112"]) < TOL
    assert golden_meta["inception_extract_112_raises"] == "RuntimeError"
    with pytest.raises(RuntimeError):  # Q4: 4x4 map < 7x7 pool
        inception_i3d_ref.extract_features(x[:1], sd_inception)


def test_same_pad_rule():
    # i3d.py:82-86 on the shapes of Appendix A.1
    assert inception_i3d_ref.same_pad(224, 7, 2) == (2, 3)
    assert inception_i3d_ref.same_pad(16, 7, 2) == (2, 3)
    assert inception_i3d_ref.same_pad(112, 3, 2) == (0, 1)
    assert inception_i3d_ref.same_pad(28, 3, 1) == (1, 1)
    assert inception_i3d_ref.same_pad(7, 2, 2) == (0, 1)
    assert inception_i3d_ref.same_pad(8, 1, 1) == (0, 0)


def test_unet(golden, golden_meta, sd_unet):
    frames = synth_tensor(SEED, "unet_frames", (4, 3, 112, 112))
    with torch.no_grad():
        y = unet_ref.forward(frames, sd_unet)
        yt = unet_ref.forward(frames, sd_unet, train=True)
    assert y.shape == (4, 3, 112, 112)
    assert rel_l2(y[0, :, 40:56, 40:56], golden["unet_out_crop"]) < TOL
    assert rel_l2(y.mean(3), golden["unet_out_rowmeans"]) < TOL
    assert abs(float(y.double().norm()) - golden_meta["unet_out_cks"][1]) < 1e-4 * golden_meta["unet_out_cks"][1]
    assert rel_l2(yt[0, :, 40:56, 40:56], golden["unet_train_out_crop"]) < 1e-4


def _unit(name, shape):
    return torch.nn.functional.normalize(synth_tensor(SEED, name, shape, -1, 1), dim=1)


def test_ntxent(golden):
    zi, zj = _unit("ntx_zi", (12, 128)), _unit("ntx_zj", (12, 128))
    v = losses_ref.nt_xent_np(zi.numpy(), zj.numpy(), 0.1)
    assert abs(v - golden["ntxent_value"][0]) < 1e-5 * abs(v)
    a, b = zi.double().requires_grad_(), zj.double().requires_grad_()
    l = losses_ref.nt_xent_torch(a, b, 0.1)
    assert abs(l.item() - v) < 1e-9 * abs(v)  # closed form == literal form
    l.backward()
    assert rel_l2(a.grad, golden["ntxent_grad_zi"]) < 1e-5
    assert rel_l2(b.grad, golden["ntxent_grad_zj"]) < 1e-5


def test_triplet_and_ce(golden):
    a, p, n = (_unit("trip_" + s, (8, 128)).double().requires_grad_() for s in "apn")
    l = losses_ref.triplet_torch(a, p, n)
    assert abs(l.item() - golden["triplet_value"][0]) < 1e-6
    assert abs(losses_ref.triplet_np(a.detach(), p.detach(), n.detach()) - l.item()) < 1e-12
    l.backward()
    for t, k in ((a, "a"), (p, "p"), (n, "n")):
        assert rel_l2(t.grad, golden["triplet_grad_" + k]) < 1e-5
    lg = synth_tensor(SEED, "ce_logits", (8, 102), -3, 3).double().requires_grad_()
    lab = (synth_tensor(SEED, "ce_labels", (8,)) * 101).long() + 1
    assert np.array_equal(lab.numpy(), golden["ce_labels"])
    lc = losses_ref.cross_entropy_torch(lg, lab)
    assert abs(lc.item() - golden["ce_value"][0]) < 1e-6
    assert abs(losses_ref.cross_entropy_np(lg.detach(), lab) - lc.item()) < 1e-12
    lc.backward()
    assert rel_l2(lg.grad, golden["ce_grad"]) < 1e-5


def test_q1_feed_and_npy_layout(golden, golden_meta, tmp_path):
    fr, co = extract_ref.q1_index_map(16, 3)
    assert np.array_equal(fr, golden["q1_frame_of"])
    assert np.array_equal(co, golden["q1_colour_of"])
    assert golden_meta["q1_fa_in_shape"] == [16, 3, 2, 2]
    assert golden_meta["q1_ft_in_shape"] == [1, 3, 16, 2, 2]
    # the same probe the reference function was run with
    T, C, H, W = 16, 3, 2, 2
    code = (torch.arange(T).view(T, 1, 1, 1) * 10 + torch.arange(C).view(1, C, 1, 1)).float().expand(T, C, H, W).contiguous()
    ft = lambda x: x[:, :, :, 0, 0].reshape(1, -1, 1, 1, 1)[:, :7]
    rows = extract_ref.extract_video([code, code + 1000], ft, fa=lambda x: x, layout="reference")
    assert rows.dtype == np.float64 and golden_meta["npy_dtype"] == "<f8"
    assert np.array_equal(rows, golden["npy_probe_rows"])
    assert golden_meta["npy_fortran"] is False


def test_process_feat_and_consumer(golden, tmp_path):
    for T in (225, 20, 33):
        ramp = synth_tensor(SEED, "mgfn_feat_%d" % T, (T, 64), -1, 1).numpy()
        assert np.allclose(extract_ref.process_feat(ramp, 32), golden["process_feat_%d" % T], atol=1e-7)
    feats = np.zeros((225, 2048))
    feats[:] = synth_tensor(SEED, "rows", (225, 2048)).numpy()
    p = str(tmp_path / "v.npy")
    np.save(p, feats)
    assert extract_ref.mgfn_getitem(p).shape == (1, 32, 2049)
    assert extract_ref.mgfn_getitem(p, test_mode=True).shape == (225, 1, 2049)
    np.save(p, np.zeros((225, 10, 2048)))
    assert extract_ref.mgfn_getitem(p).shape == (10, 32, 2049)


def test_train_step_oracle_vs_reference_modules(golden_meta, sd_largei3d, sd_unet):
    """oracle/train_step_ref.py against loss values + per-parameter gradient norms obtained by running the
    reference modules through the loss lines of train_anonymizer.py (make_golden.py g7)."""
    from oracle import train_step_ref
    g = golden_meta["train_step"]
    video = synth_train_video(SEED, "train_video", (2, 48, 3, 32, 32))
    labels = torch.tensor([5, 77])
    l1, grads1, _ = train_step_ref.phase1(video, labels, sd_unet, sd_largei3d)
    assert abs(l1["loss_fa"] - g["phase1"]["loss_fa"]) < 2e-4 * abs(g["phase1"]["loss_fa"])
    for k, ref in g["phase1"]["grad_l2"].items():
        # conv biases in front of a train-mode BN have an analytically ZERO gradient (noise ~1e-6 both sides)
        assert abs(float(grads1[k].norm()) - ref) <= 5e-3 * ref + 2e-5, k
    video64 = synth_train_video(SEED, "train_video64", (4, 48, 3, 64, 64))
    l2, grads2 = train_step_ref.phase2(video64, torch.tensor([5, 77, 101, 1]), sd_unet, sd_largei3d)
    assert abs(l2["loss_ft"] - g["phase2"]["loss_ft"]) < 2e-4 * abs(g["phase2"]["loss_ft"])
    bad = [k for k, ref in g["phase2"]["grad_l2"].items() if abs(float(grads2[k].norm()) - ref) > 2e-2 * ref + 1e-6]
    assert len(bad) <= 3, bad   # a handful of tiny-norm BN gradients differ in fp32 summation order
    assert g["phase2"]["num_batches_tracked"] == 3   # Q14: three train-mode forwards per step


def test_checkpointed_unet_oracle_gives_the_same_gradients(sd_largei3d, sd_unet):
    """oracle/unet_ref.forward(checkpoint=True) (what lets the full cfg3 batch of phase 1 fit the host in tests/test_hip_train_golden.py) is the same function:
    loss and every fa gradient bit-equal to the plain path."""
    from oracle import train_step_ref
    video = synth_train_video(SEED, "train_video", (2, 48, 3, 32, 32))
    labels = torch.tensor([5, 77])
    a = train_step_ref.phase1(video, labels, sd_unet, sd_largei3d)
    b = train_step_ref.phase1(video, labels, sd_unet, sd_largei3d, checkpoint=True)
    assert a[0] == b[0] and set(a[1]) == set(b[1])
    assert all(torch.equal(a[1][k], b[1][k]) for k in a[1])


def test_train_step_phase1_with_the_privacy_term_vs_reference_modules(golden_meta, sd_largei3d, sd_unet):
    """Phase 1 WITH the NT-Xent term (train_anonymizer.py:73-84,119): golden g7 `phase1_fb` ran the reference's UNet / I3Res50 modules and its own
    NTXentLoss class through those lines with a small conv net standing in for fb (make_golden.py; the real ResNet-50 is torchvision's). The oracle with the
    same stand-in: loss_fa = -1.0 * NTXent + 0.7 * loss_ft, every term and the gradient norm of every fa parameter."""
    import torch.nn.functional as F
    from oracle import train_step_ref
    from ted_spad_amd.synth import synth_tensor
    g = golden_meta["train_step"]["phase1_fb"]
    cw, cb = synth_tensor(SEED, "stubfb.conv.weight", (8, 3, 3, 3), -0.5, 0.5), synth_tensor(SEED, "stubfb.conv.bias", (8,), -0.1, 0.1)
    fw, fbias = synth_tensor(SEED, "stubfb.fc.weight", (128, 8), -1, 1), synth_tensor(SEED, "stubfb.fc.bias", (128,), -0.1, 0.1)
    fb = lambda x: F.normalize(F.linear(F.relu(F.conv2d(x, cw, cb, stride=2, padding=1)).mean((2, 3)), fw, fbias), p=2, dim=1)
    views = [synth_tensor(SEED, "vispr_view%d" % v, (4, 3, 32, 32)) for v in range(2)]
    video = synth_train_video(SEED, "train_video", (2, 48, 3, 32, 32))
    l1, grads, _ = train_step_ref.phase1(video, torch.tensor([5, 77]), sd_unet, sd_largei3d, vispr=views, fb_fn=fb)
    for k in ("loss_fa", "loss_ft", "loss_fb"):
        assert abs(l1[k] - g[k]) < 3e-4 * abs(g[k]), (k, l1[k], g[k])
    assert abs(l1["loss_fa"] - (-1.0 * l1["loss_fb"] + 0.7 * l1["loss_ft"])) < 1e-5
    for k, ref in g["grad_l2"].items():
        assert abs(float(grads[k].norm()) - ref) <= 5e-3 * ref + 2e-5, k
    assert g["num_batches_tracked"] == 3        # Q14: fa's BatchNorms saw the two views and the video's pseudo-images


@pytest.mark.parametrize("hw", [(480, 856), (97, 131), (224, 224)])
def test_pil_resample_tables_reproduce_pillow(hw):
    """Host logic of the shanghai pre-processing path: `preprocess.pil_table` (libImaging/Resample.c restated) driving the two-pass
    integer resample in numpy must equal Pillow's own Image.resize(BILINEAR) -- through the oracle, which calls Pillow."""
    import numpy as np
    from oracle import preprocess_ref
    from ted_spad_amd.preprocess import pil_table
    h, w = hw
    img = np.random.default_rng(h).integers(0, 256, (h, w, 3), dtype=np.uint8)
    ref = preprocess_ref.shanghai_augmentation(img).numpy()
    side = int(h * 0.8)
    top, left = int(round((h - side) / 2.0)), int(round((w - side) / 2.0))
    crop = img[top:top + side, left:left + side].astype(np.int64)
    tab, _ = pil_table(side, 224)
    tmp = np.zeros((side, 224, 3), np.int64)
    for ox in range(224):
        x0, n = tab[ox, 0], tab[ox, 1]
        tmp[:, ox] = np.clip(((1 << 21) + (crop[:, x0:x0 + n] * tab[ox, 2:2 + n].astype(np.int64)[None, :, None]).sum(1)) >> 22, 0, 255)
    out = np.zeros((224, 224, 3), np.int64)
    for oy in range(224):
        y0, n = tab[oy, 0], tab[oy, 1]
        out[oy] = np.clip(((1 << 21) + (tmp[y0:y0 + n] * tab[oy, 2:2 + n].astype(np.int64)[:, None, None]).sum(0)) >> 22, 0, 255)
    got = (out.astype(np.float32) / np.float32(255)).transpose(2, 0, 1)
    assert np.array_equal(got, ref)


def test_gradient_sensitivity_to_f16_activations(sd_largei3d):
    """How well-conditioned is the quantity the GPU gradient tests compare against? The fp32 oracle trunk in train mode (batch
    statistics), cross-entropy on the pooled feature, gradients w.r.t. all trunk parameters -- once in fp32, once with every
    activation rounded to f16 in the FORWARD only (straight-through: the backward arithmetic stays fp32). On the randomly
    initialised network the gradients move by tens of percent (ReLU decisions flipping next to 0, BatchNorm's backward subtracting
    two nearly equal terms): the loose end-to-end gradient bounds of tests/test_hip_train_step.py / test_hip_train_golden.py are in
    proportion to this, not a licence for wrong kernels (those are held to 1e-3 .. 5e-3 one by one, and 0.4-1.6 % on the smooth chains)."""
    from synth_helpers import rounded_forward_gradients
    l0, l1, errs, cos = rounded_forward_gradients(sd_largei3d)
    assert abs(l0 - l1) < 2e-3 * abs(l0)                          # the loss barely moves ...
    med = float(np.median(errs))
    assert 0.05 < med < 0.6 and float(np.median(cos)) > 0.9, (med, float(np.median(cos)))   # ... the gradients do (measured: 0.19 .. 0.28)
