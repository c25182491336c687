"""Host time from the start of a training phase to its first C-ABI launch, and from its last launch to its return (the stretches a per-phase loss read-back
leaves the GPU idle for). Usage: python scripts/phase_prologue_probe.py"""
import os, sys, io, contextlib, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ted_spad_amd import engine as E, _lib
from ted_spad_amd.model_loaders import load_fa_model, load_fb_model, load_ft_model
from ted_spad_amd.synth import synth_state_dict, synth_tensor, synth_train_video
from ted_spad_amd.train_step import AnonymizerTrainStep
with contextlib.redirect_stdout(io.StringIO()):
    fa, ft = load_fa_model(arch="unet"), load_ft_model("largei3d", num_classes=102)
    fb = load_fb_model(arch="r50", ssl=True, pretrained=False)
for m in (fa, ft, fb):
    m.load_state_dict(synth_state_dict(m.state_dict(), 0))
video = synth_train_video(0, "v", (8, 48, 3, 112, 112), device="cuda"); labels = torch.randint(1, 102, (8,), device="cuda")
views = [synth_tensor(0, "vispr_view%d" % v, (12, 3, 224, 224), device="cuda") for v in range(2)]
step = AnonymizerTrainStep(fa.cuda(), ft.cuda(), fb_model=fb.cuda())
fns = (("phase 1", lambda: step.step_fa(video, labels, views)), ("phase 2", lambda: step.step_ft(video, labels, inputs_vispr=views)))
for _, fn in fns:
    for i in range(180):
        if i >= 45 and not E.tuning_pending():
            break
        fn()
L = _lib.lib()
stamps = []
for nm in _lib.SYMBOLS:
    f = getattr(L, nm)
    def wrap(f=f):
        def g(*a):
            stamps.append(time.perf_counter())
            return f(*a)
        return g
    setattr(L, nm, wrap())
for name, fn in fns:
    pro, epi, tot = [], [], []
    for _ in range(10):
        torch.cuda.synchronize(); stamps.clear()
        t0 = time.perf_counter(); fn(); t1 = time.perf_counter()
        pro.append(stamps[0] - t0); epi.append(t1 - stamps[-1]); tot.append(t1 - t0)
    med = lambda v: sorted(v)[len(v) // 2] * 1e3
    print("%s: %.2f ms from the call to the first launch, %.2f ms from the last launch to the return (incl. the loss read-back), %.2f ms in all, %d launches" % (name, med(pro), med(epi), med(tot), len(stamps)))
