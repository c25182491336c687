import os, sys, io, contextlib, warnings, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ted_spad_amd.model_loaders import load_fa_model, load_ft_model
from ted_spad_amd.synth import synth_state_dict, synth_train_video
from ted_spad_amd.train_step import AnonymizerTrainStep
with contextlib.redirect_stdout(io.StringIO()):
    fa, ft = load_fa_model(arch="unet"), load_ft_model("largei3d", num_classes=102)
fa.load_state_dict(synth_state_dict(fa.state_dict(), 0)); ft.load_state_dict(synth_state_dict(ft.state_dict(), 0))
step = AnonymizerTrainStep(fa.cuda(), ft.cuda()); step.lazy_losses = True
video = synth_train_video(0, "v", (8, 48, 3, 112, 112), device="cuda"); labels = torch.randint(1, 102, (8,), device="cuda")
os.environ["TEDSPAD_AUTOTUNE"] = "0"
from ted_spad_amd import engine as E
E.AUTOTUNE = False
for _ in range(3): step.step_fa(video, labels); step.step_ft(video, labels)
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    step.step_fa(video, labels); step.step_ft(video, labels)
torch.cuda.set_sync_debug_mode("default")
print("synchronizing calls inside one lazy iteration:", len(w))
for x in w[:10]:
    print("  ", str(x.message)[:100], "@", x.filename.split("/")[-1], x.lineno)
