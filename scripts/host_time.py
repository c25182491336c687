"""Host (Python + ctypes) time of a training phase against its GPU time: lazy losses, no sync inside the loop; the host time is the wall time of the
calls alone, the GPU time the wall time including the final synchronize. Usage: python scripts/host_time.py [--fb]   (HOST_PROFILE=1: cProfile of both phases; COUNT_CALLS=1: C-ABI calls per phase)"""
import os, sys, io, contextlib, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ted_spad_amd import engine as E
from ted_spad_amd.model_loaders import load_fa_model, load_fb_model, load_ft_model
from ted_spad_amd.synth import synth_state_dict, synth_tensor, synth_train_video
from ted_spad_amd.train_step import AnonymizerTrainStep
with contextlib.redirect_stdout(io.StringIO()):
    fa, ft = load_fa_model(arch="unet"), load_ft_model("largei3d", num_classes=102)
fa.load_state_dict(synth_state_dict(fa.state_dict(), 0)); ft.load_state_dict(synth_state_dict(ft.state_dict(), 0))
FB = "--fb" in sys.argv          # the whole train_epoch body: privacy branch fb (ResNet-50 + MLP) + NT-Xent on the two VISPR views (bench.py's train_cfg3)
video = synth_train_video(0, "v", (8, 48, 3, 112, 112), device="cuda"); labels = torch.randint(1, 102, (8,), device="cuda")
if FB:
    with contextlib.redirect_stdout(io.StringIO()):
        fb = load_fb_model(arch="r50", ssl=True, pretrained=False)
    fb.load_state_dict(synth_state_dict(fb.state_dict(), 0))
    views = [synth_tensor(0, "vispr_view%d" % v, (12, 3, 224, 224), device="cuda") for v in range(2)]
    step = AnonymizerTrainStep(fa.cuda(), ft.cuda(), fb_model=fb.cuda()); step.lazy_losses = True
    _fa, _ft = step.step_fa, step.step_ft
    step_fa = lambda v, l: _fa(v, l, views)
    step_ft = lambda v, l: _ft(v, l, inputs_vispr=views)
else:
    step = AnonymizerTrainStep(fa.cuda(), ft.cuda()); step.lazy_losses = True
    step_fa, step_ft = step.step_fa, step.step_ft
for fn in (step_fa, step_ft):
    for i in range(180):
        if i >= 45 and not E.tuning_pending():
            break
        fn(video, labels)
for name, fn in (("phase 1", step_fa), ("phase 2", step_ft)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn(video, labels)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s: host %.2f ms per step, with the GPU %.2f ms per step" % (name, (t1 - t0) * 100, (t2 - t0) * 100))
if os.environ.get("COUNT_CALLS"):
    from ted_spad_amd import _lib
    L = _lib.lib()
    counts = {}
    for nm in _lib.SYMBOLS:
        f = getattr(L, nm)
        def wrap(f=f, nm=nm):
            def g(*a):
                counts[nm] = counts.get(nm, 0) + 1
                return f(*a)
            return g
        setattr(L, nm, wrap())
    for name, fn in (("phase 1", step_fa), ("phase 2", step_ft)):
        counts.clear(); fn(video, labels); torch.cuda.synchronize()
        print("%s: %d C-ABI calls: %s" % (name, sum(counts.values()), sorted(counts.items(), key=lambda t: -t[1])))
if os.environ.get("HOST_PROFILE"):
    import cProfile, pstats
    for name, fn in (("phase 2", step_ft), ("phase 1", step_fa)):
        pr = cProfile.Profile(); pr.enable()
        for _ in range(10):
            fn(video, labels)
        pr.disable(); torch.cuda.synchronize()
        print("----", name)
        pstats.Stats(pr).sort_stats("tottime").print_stats(22)
