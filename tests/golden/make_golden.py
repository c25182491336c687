"""Generate the golden fixtures by RUNNING THE REFERENCE in the authoring container.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz / *.json

The reference's Python files are imported in place from /root/reference (never copied);
only inputs-by-recipe and expected OUTPUTS are committed. Weights and inputs are not
stored: they are regenerated anywhere from `ted_spad_amd.synth` (portable counter-based
generator), keyed by (seed, tensor name).

What is captured:
  g1  I3Res50.extract_features : 2 clips 16x224^2 + the cfg1 clip 16x112^2 (+ per-stage checksums)
  g2  InceptionI3d.extract_features : 2 clips 16x224^2 ; InceptionI3d.forward logits @112^2
  g4  UNet.forward : 4 frames @112^2 (checksums + one 16x16 crop)
  g5  wrapper_i3d.forward (pred, feat), B=2 @112^2, eval and train mode (dropout p=0)
  g6  NTXentLoss / TripletMarginLoss / CrossEntropyLoss values + input gradients
  g8  Q1 feed index map + `.npy` header, by calling the reference's own
      st_feature_extraction.extract_features with probe models
  g9  MGFN utils.process_feat on a (T,2048) ramp
"""
import io
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

from _refimport import REFERENCE_ROOT, import_reference, _stub  # noqa: E402
from ted_spad_amd.synth import synth_clips, synth_state_dict, synth_tensor, synth_train_video  # noqa: E402

SEED = 0


def cks(t):
    t = t.detach().double()
    return [float(t.mean()), float(t.norm())]


def hook_taps(model, names, store):
    hs = []
    mods = dict(model.named_modules())
    for n in names:
        hs.append(mods[n].register_forward_hook(lambda m, i, o, n=n: store.__setitem__(n, cks(o))))
    return hs


def main():
    torch.set_num_threads(os.cpu_count())
    ml, NTXentLoss = import_reference()
    meta = {"seed": SEED, "torch": torch.__version__, "reference": REFERENCE_ROOT}
    out = {}

    # ---------------- g1 / g5 : largei3d -------------------------------------------------
    ft = ml.load_ft_model("largei3d", num_classes=102, kin_pretrained=False)
    sd = synth_state_dict(ft.state_dict(), SEED)
    ft.load_state_dict(sd, strict=True)
    ft.eval()
    taps = {}
    hs = hook_taps(ft.i3d, ["relu", "maxpool1", "layer1", "layer2", "layer3", "layer4"], taps)
    # NB: `relu` is shared by every block; only its FIRST call (the stem) is kept.
    first = {}
    def _stem_hook(m, i, o):
        first.setdefault("stem", cks(o))

    hs.append(ft.i3d.relu.register_forward_hook(_stem_hook))
    with torch.no_grad():
        x224 = synth_clips(SEED, 2, (3, 16, 224, 224))
        f224 = ft.i3d.extract_features(x224)
        t224 = dict(taps, stem=first.pop("stem"))
        x112 = synth_clips(SEED, 1, (3, 16, 112, 112))
        f112 = ft.i3d.extract_features(x112)
    for h in hs:
        h.remove()
    t224.pop("relu", None)
    out["i3res50_feat_224"] = f224.reshape(2, 2048).numpy()
    out["i3res50_feat_112"] = f112.reshape(1, 2048).numpy()
    meta["i3res50_taps_224"] = t224

    ft.i3d.drop.p = 0.0  # Q13: dropout made deterministic for the fixture
    x2 = synth_clips(SEED, 2, (3, 16, 112, 112))
    with torch.no_grad():
        pred_e, feat_e = ft(x2)
    out["wrapper_eval_pred"] = pred_e.numpy()
    out["wrapper_eval_feat"] = feat_e.float().numpy()
    ft.train()
    pred_t, feat_t = ft(x2)
    out["wrapper_train_pred"] = pred_t.detach().numpy()
    out["wrapper_train_feat"] = feat_t.detach().float().numpy()
    sd_after = ft.state_dict()
    meta["wrapper_train_running_stats"] = {
        k: cks(sd_after[k]) for k in ("i3d.bn1.running_mean", "i3d.bn1.running_var",
                                      "i3d.layer4.2.bn3.running_mean", "i3d.layer4.2.bn3.running_var",
                                      "mlp.bn1.running_mean", "mlp.bn2.running_var")}
    meta["wrapper_train_num_batches_tracked"] = int(sd_after["i3d.bn1.num_batches_tracked"])
    meta["wrapper_state_dict_keys"] = len(sd_after)
    meta["wrapper_params"] = int(sum(p.numel() for p in ft.parameters()))
    del ft

    # ---------------- g2 : inception i3d --------------------------------------------------
    inc = ml.load_ft_model("i3d", num_classes=102, kin_pretrained=False)
    sdi = synth_state_dict(inc.state_dict(), SEED)
    inc.load_state_dict(sdi, strict=True)
    inc.eval()
    taps = {}
    names = [n for n in inc.VALID_ENDPOINTS if n in inc.end_points]
    hs = hook_taps(inc, names, taps)
    with torch.no_grad():
        fi = inc.extract_features(x224)
        ti = dict(taps)
        li = inc(synth_clips(SEED, 2, (3, 16, 112, 112)))
    for h in hs:
        h.remove()
    out["inception_feat_224"] = fi.reshape(2, 1024).numpy()
    out["inception_logits_112"] = li.numpy()
    meta["inception_taps_224"] = ti
    meta["inception_state_dict_keys"] = len(sdi)
    meta["inception_params"] = int(sum(p.numel() for p in inc.parameters()))
    try:
        inc.extract_features(synth_clips(SEED, 1, (3, 16, 112, 112)))
        meta["inception_extract_112_raises"] = False
    except Exception as e:  # Q4
        meta["inception_extract_112_raises"] = type(e).__name__
    del inc

    # ---------------- g4 : UNet ---------------------------------------------------------------
    fa = ml.load_fa_model(arch="unet")
    sda = synth_state_dict(fa.state_dict(), SEED)
    fa.load_state_dict(sda, strict=True)
    fa.eval()
    frames = synth_tensor(SEED, "unet_frames", (4, 3, 112, 112))
    with torch.no_grad():
        y = fa(frames)
    out["unet_out_crop"] = y[0, :, 40:56, 40:56].numpy()
    out["unet_out_rowmeans"] = y.mean(dim=3).numpy()  # (4,3,112)
    meta["unet_out_cks"] = cks(y)
    meta["unet_state_dict_keys"] = len(sda)
    meta["unet_params"] = int(sum(p.numel() for p in fa.parameters()))
    fa.train()
    yt = fa(frames)
    meta["unet_train_out_cks"] = cks(yt)
    out["unet_train_out_crop"] = yt[0, :, 40:56, 40:56].detach().numpy()

    # ---------------- g6 : losses ----------------------------------------------------------------
    def unit(name, shape):
        return torch.nn.functional.normalize(synth_tensor(SEED, name, shape, -1, 1), dim=1)

    zi = unit("ntx_zi", (12, 128)).requires_grad_()
    zj = unit("ntx_zj", (12, 128)).requires_grad_()
    ntx = NTXentLoss(device="cpu", batch_size=12, temperature=0.1, use_cosine_similarity=False)
    l = ntx(zi, zj)
    l.backward()
    out["ntxent_value"] = np.array([l.item()])
    out["ntxent_grad_zi"] = zi.grad.numpy()
    out["ntxent_grad_zj"] = zj.grad.numpy()
    a, p, n = (unit("trip_" + s, (8, 128)).requires_grad_() for s in "apn")
    lt = torch.nn.TripletMarginLoss(margin=1)(a, p, n)
    lt.backward()
    out["triplet_value"] = np.array([lt.item()])
    out["triplet_grad_a"], out["triplet_grad_p"], out["triplet_grad_n"] = a.grad.numpy(), p.grad.numpy(), n.grad.numpy()
    lg = synth_tensor(SEED, "ce_logits", (8, 102), -3, 3).requires_grad_()
    lab = (synth_tensor(SEED, "ce_labels", (8,)) * 101).long() + 1
    lc = torch.nn.CrossEntropyLoss()(lg, lab)
    lc.backward()
    out["ce_value"] = np.array([lc.item()])
    out["ce_grad"] = lg.grad.numpy()
    out["ce_labels"] = lab.numpy()

    # ---------------- g8 : the reference's own extract_features with probe models ---------------
    _stub("cv2")
    tvt = _stub("torchvision.transforms")
    sys.modules["torchvision"].transforms = tvt
    fe_dir = os.path.join(REFERENCE_ROOT, "feature_extraction")
    sys.path.insert(0, fe_dir)
    cwd = os.getcwd()
    os.chdir(fe_dir)
    try:
        import st_feature_extraction as stfe
    finally:
        os.chdir(cwd)
    torch.Tensor.cuda = lambda self, *a, **k: self  # the function hard-codes .cuda() (:18)

    seen = {}

    class ProbeFa(torch.nn.Module):
        def forward(self, x):
            seen["fa_in_shape"] = list(x.shape)
            return x

    class ProbeFt:
        def extract_features(self, x):
            seen["ft_in_shape"] = list(x.shape)
            seen["ft_in"] = x.clone()
            return x[:, :, :, 0, 0].reshape(1, -1, 1, 1, 1)[:, :7]

    T, C, H, W = 16, 3, 2, 2
    code = (torch.arange(T).view(T, 1, 1, 1) * 10 + torch.arange(C).view(1, C, 1, 1)).float().expand(T, C, H, W).contiguous()
    vid = [code, code + 1000]
    feats = np.zeros((len(vid), 7))
    with tempfile.TemporaryDirectory() as td:
        pth = os.path.join(td, "probe.npy")
        stfe.extract_features(vid, feats, pth, ProbeFa(), ProbeFt(), True, False)
        raw = open(pth, "rb").read()
        arr = np.load(io.BytesIO(raw))
    ftin = seen["ft_in"][0, :, :, 0, 0] - 1000  # (3,16) codes of the 2nd clip
    out["q1_frame_of"] = (ftin // 10).numpy().astype(np.int64)
    out["q1_colour_of"] = (ftin % 10).numpy().astype(np.int64)
    meta["q1_fa_in_shape"] = seen["fa_in_shape"]
    meta["q1_ft_in_shape"] = seen["ft_in_shape"]
    meta["npy_dtype"] = arr.dtype.str
    meta["npy_shape"] = list(arr.shape)
    meta["npy_fortran"] = bool(np.isfortran(arr))
    out["npy_probe_rows"] = arr

    # ---------------- g9 : MGFN process_feat -------------------------------------------------
    _stub("visdom")
    mg = os.path.join(REFERENCE_ROOT, "anomaly_detection_mgfn")
    sys.path.insert(0, mg)
    mgu = types.ModuleType("mgfn_utils")
    src = os.path.join(mg, "utils", "utils.py")
    import importlib.util
    spec = importlib.util.spec_from_file_location("mgfn_utils", src)
    mgu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mgu)
    for Tn in (225, 20, 33):
        ramp = synth_tensor(SEED, "mgfn_feat_%d" % Tn, (Tn, 64), -1, 1).numpy()
        out["process_feat_%d" % Tn] = mgu.process_feat(ramp, 32)

    np.savez_compressed(os.path.join(HERE, "golden.npz"), **out)
    with open(os.path.join(HERE, "golden_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", {k: v.shape for k, v in out.items()})


if __name__ == "__main__" and "--train-step" not in sys.argv:
    main()
    sys.argv.append("--train-step")


def train_step_golden():
    """g7: the REFERENCE modules through the loss lines of train_anonymizer.py (phase 1 :87-123, phase 2 :137-191)
    at tiny size (B=2, 48 frames of 32x32), fb term absent (torchvision ResNet-50 is not available). Appended to
    golden.npz / golden_meta.json."""
    ml, _ = import_reference()
    torch.set_num_threads(os.cpu_count())
    fa = ml.load_fa_model(arch="unet")
    ft = ml.load_ft_model("largei3d", num_classes=102, kin_pretrained=False)
    fa.load_state_dict(synth_state_dict(fa.state_dict(), SEED))
    ft.load_state_dict(synth_state_dict(ft.state_dict(), SEED))
    ft.i3d.drop.p = 0.0
    video = synth_train_video(SEED, "train_video", (2, 48, 3, 32, 32))
    labels = torch.tensor([5, 77])
    crit, trip = torch.nn.CrossEntropyLoss(), torch.nn.TripletMarginLoss(margin=1)
    meta = {}
    # ---- phase 1 (train_anonymizer.py:71-123) ----
    fa.train(); ft.eval()
    iv = video.permute(0, 2, 1, 3, 4)
    ori = iv.shape
    anon = fa(iv.reshape(-1, ori[1], ori[3], ori[4])).reshape(ori)
    i1, i2, i3 = torch.split(anon, [16, 16, 16], dim=2)
    out, f1 = ft(i1); _, f2 = ft(i2); _, f3 = ft(i3)
    loss_ft = crit(out, labels) + 0.1 * trip(f1, f2, f3)
    loss_fa = 0.7 * loss_ft
    loss_fa.backward()
    meta["phase1"] = dict(loss_fa=loss_fa.item(), loss_ft=loss_ft.item(),
                          grad_l2={k: float(p.grad.norm()) for k, p in fa.named_parameters()})
    # ---- phase 2 (train_anonymizer.py:135-191) on fresh modules; 64x64 so the last stage still has 2x2x2 positions
    #      (at 32x32 layer4's train-mode BN would normalise over 2 values: analytically zero gradients, pure noise) ----
    fa.load_state_dict(synth_state_dict(fa.state_dict(), SEED)); fa.zero_grad(); ft.zero_grad()
    fa.eval(); ft.train()
    # B = 4: with B = 2 the mlp's train-mode BatchNorm1d normalises over two values (analytically zero gradient).
    video = synth_train_video(SEED, "train_video64", (4, 48, 3, 64, 64))
    labels = torch.tensor([5, 77, 101, 1])
    iv = video.permute(0, 2, 1, 3, 4)
    ori = iv.shape
    with torch.no_grad():
        anon = fa(iv.reshape(-1, ori[1], ori[3], ori[4])).reshape(ori)
    i1, i2, i3 = torch.split(anon, [16, 16, 16], dim=2)
    out, f1 = ft(i1); _, f2 = ft(i2); _, f3 = ft(i3)
    loss_ft = crit(out, labels) + 0.1 * trip(f1, f2, f3)
    loss_ft.backward()
    meta["phase2"] = dict(loss_ft=loss_ft.item(), grad_l2={k: float(p.grad.norm()) for k, p in ft.named_parameters()},
                          num_batches_tracked=int(ft.i3d.bn1.num_batches_tracked))
    # ---- phase 1 WITH the privacy term (train_anonymizer.py:73-84,119): the reference's own NTXentLoss inside a golden step. The real fb (torchvision ResNet-50)
    #      is not importable here; a small conv net defined HERE stands in for it (frozen, eval): what is pinned is the loss algebra
    #      -fb_loss_weight * NTXent(fb(fa(v0)), fb(fa(v1))) + ft_loss_weight * loss_ft, the two extra train-mode fa forwards and the gradient flow into fa ----
    from aux_code.nt_xent_original import NTXentLoss
    fa.load_state_dict(synth_state_dict(fa.state_dict(), SEED)); fa.zero_grad(); ft.zero_grad()
    ft.load_state_dict(synth_state_dict(ft.state_dict(), SEED))
    fa.train(); ft.eval()
    stub = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, stride=2, padding=1), torch.nn.ReLU(), torch.nn.AdaptiveAvgPool2d(1), torch.nn.Flatten(),
                               torch.nn.Linear(8, 128))
    with torch.no_grad():
        stub[0].weight.copy_(synth_tensor(SEED, "stubfb.conv.weight", (8, 3, 3, 3), -0.5, 0.5)); stub[0].bias.copy_(synth_tensor(SEED, "stubfb.conv.bias", (8,), -0.1, 0.1))
        stub[4].weight.copy_(synth_tensor(SEED, "stubfb.fc.weight", (128, 8), -1, 1)); stub[4].bias.copy_(synth_tensor(SEED, "stubfb.fc.bias", (128,), -0.1, 0.1))
    stub.eval()
    fb = lambda x: torch.nn.functional.normalize(stub(x), p=2, dim=1)
    views = [synth_tensor(SEED, "vispr_view%d" % v, (4, 3, 32, 32)) for v in range(2)]
    video = synth_train_video(SEED, "train_video", (2, 48, 3, 32, 32))
    labels = torch.tensor([5, 77])
    output1 = [fb(fa(v)) for v in views]                                  # :80
    loss_fb = NTXentLoss("cpu", 4, 0.1, False)(output1[0], output1[1])    # :82-84
    iv = video.permute(0, 2, 1, 3, 4)
    ori = iv.shape
    anon = fa(iv.reshape(-1, ori[1], ori[3], ori[4])).reshape(ori)
    i1, i2, i3 = torch.split(anon, [16, 16, 16], dim=2)
    out, f1 = ft(i1); _, f2 = ft(i2); _, f3 = ft(i3)
    loss_ft = crit(out, labels) + 0.1 * trip(f1, f2, f3)
    loss_fa = -1.0 * loss_fb + 0.7 * loss_ft                              # :119
    loss_fa.backward()
    meta["phase1_fb"] = dict(loss_fa=loss_fa.item(), loss_ft=loss_ft.item(), loss_fb=loss_fb.item(),
                             grad_l2={k: float(p.grad.norm()) for k, p in fa.named_parameters()},
                             num_batches_tracked=int(fa.inc.double_conv[1].num_batches_tracked))
    path = os.path.join(HERE, "golden_meta.json")
    full = json.load(open(path))
    full["train_step"] = meta
    json.dump(full, open(path, "w"), indent=1, sort_keys=True)
    print("train-step golden:", meta["phase1"]["loss_fa"], meta["phase2"]["loss_ft"])


if __name__ == "__main__" and "--train-step" in sys.argv:
    train_step_golden()
