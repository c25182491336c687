"""Host (Python + ctypes) time of a training phase against its GPU time: lazy losses, no sync inside the loop; the host time is the wall time of the
calls alone, the GPU time the wall time including the final synchronize. Usage: python scripts/host_time.py"""
import os, sys, io, contextlib, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ted_spad_amd import engine as E
from ted_spad_amd.model_loaders import load_fa_model, load_ft_model
from ted_spad_amd.synth import synth_state_dict, synth_train_video
from ted_spad_amd.train_step import AnonymizerTrainStep
with contextlib.redirect_stdout(io.StringIO()):
    fa, ft = load_fa_model(arch="unet"), load_ft_model("largei3d", num_classes=102)
fa.load_state_dict(synth_state_dict(fa.state_dict(), 0)); ft.load_state_dict(synth_state_dict(ft.state_dict(), 0))
step = AnonymizerTrainStep(fa.cuda(), ft.cuda()); step.lazy_losses = True
video = synth_train_video(0, "v", (8, 48, 3, 112, 112), device="cuda"); labels = torch.randint(1, 102, (8,), device="cuda")
for fn in (step.step_fa, step.step_ft):
    for i in range(180):
        if i >= 45 and not E.tuning_pending():
            break
        fn(video, labels)
for name, fn in (("phase 1", step.step_fa), ("phase 2", step.step_ft)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn(video, labels)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s: host %.2f ms per step, with the GPU %.2f ms per step" % (name, (t1 - t0) * 100, (t2 - t0) * 100))
if os.environ.get("HOST_PROFILE"):
    import cProfile, pstats
    for name, fn in (("phase 2", step.step_ft), ("phase 1", step.step_fa)):
        pr = cProfile.Profile(); pr.enable()
        for _ in range(10):
            fn(video, labels)
        pr.disable(); torch.cuda.synchronize()
        print("----", name)
        pstats.Stats(pr).sort_stats("tottime").print_stats(22)
