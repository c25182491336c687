"""Time of the in-place weight refresh (tedspad_fold_multi + tedspad_pack_multi) of the three cfg3 networks after their images exist: ms per refresh, HIP events.
Usage: python scripts/pack_probe.py"""
import os, sys, io, contextlib, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ted_spad_amd import train_engine as TE
from ted_spad_amd.model_loaders import load_fa_model, load_fb_model, load_ft_model
from ted_spad_amd.synth import synth_state_dict, synth_tensor, synth_train_video
from ted_spad_amd.train_step import AnonymizerTrainStep
with contextlib.redirect_stdout(io.StringIO()):
    fa, ft, fb = load_fa_model(arch="unet"), load_ft_model("largei3d", num_classes=102), load_fb_model(arch="r50", ssl=True, pretrained=False)
for m in (fa, ft, fb):
    m.load_state_dict(synth_state_dict(m.state_dict(), 0))
video = synth_train_video(0, "v", (2, 48, 3, 64, 64), device="cuda"); labels = torch.randint(1, 102, (2,), device="cuda")
views = [synth_tensor(0, "vispr_view%d" % v, (4, 3, 64, 64), device="cuda") for v in range(2)]
step = AnonymizerTrainStep(fa.cuda(), ft.cuda(), fb_model=fb.cuda())
for _ in range(2):                                   # both phases once: every forward / data-gradient image exists
    step.step_fa(video, labels, views); step.step_ft(video, labels, inputs_vispr=views)
torch.cuda.synchronize()
for name, tr, mod in (("fa (UNet)", step.fa_tr, fa), ("ft (I3Res50)", step.ft_tr, ft), ("fb (ResNet-50 + MLP)", step.fb_tr, fb)):
    R = tr.refresh if hasattr(tr, "refresh") else tr.trunk.refresh
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ms = []
    for i in range(12):
        TE.mark_updated(mod.parameters())
        torch.cuda.synchronize()
        ev[0].record(); did = R.run(); ev[1].record(); torch.cuda.synchronize()
        if i >= 2:
            ms.append(ev[0].elapsed_time(ev[1]))
    ms.sort()
    print("%-22s refresh %s: median %.3f ms, min %.3f ms" % (name, did, ms[len(ms) // 2], ms[0]))
