"""Anonymised extraction (dali_extraction.py:151-182 with params.anonymized = True): clips -> fa (UNet, eval) with the
reference's reshape feed (Q1) -> I3Res50.extract_features. Prints clips/s and the algorithmic TFLOP/s
(UNet 61.232 GFLOP per 224^2 frame x 16 frames + 32.829 GFLOP per clip)."""
import argparse, contextlib, io, json, os, sys, time
import torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ted_spad_amd import extraction
from ted_spad_amd.model_loaders import load_fa_model, load_ft_model
from ted_spad_amd.synth import synth_clips, synth_state_dict

ap = argparse.ArgumentParser()
ap.add_argument('--clips', type=int, default=300); ap.add_argument('--batch', type=int, default=25)
ap.add_argument('--steps', type=int, default=3); ap.add_argument('--warmup', type=int, default=4); ap.add_argument('--ft-batch', type=int, default=75)
ap.add_argument('--arch-fa', default='unet', choices=['unet', 'unet++'], help="the anonymizer: 'unet' or the reference's default 'unet++'")
a = ap.parse_args()
with contextlib.redirect_stdout(io.StringIO()):
    fa, ft = load_fa_model(arch=a.arch_fa), load_ft_model('largei3d', num_classes=102)
fa.load_state_dict(synth_state_dict(fa.state_dict(), 0)); ft.load_state_dict(synth_state_dict(ft.state_dict(), 0))
fa, ft = fa.cuda().eval(), ft.cuda().eval()
clips = synth_clips(0, a.clips, (3, 16, 224, 224), device='cuda').view(a.clips, 16, 3, 224, 224)   # loader layout (B,16,3,H,W)
out = torch.empty((a.clips, 2048), device='cuda')

def step():
    with torch.no_grad():
        for i in range(0, a.clips, a.ft_batch):
            x = extraction.feed(clips[i:i + a.ft_batch], fa, 'reference', fa_batch=a.batch)
            out[i:i + a.ft_batch] = ft.i3d.extract_features(x).flatten(1)
from ted_spad_amd import engine as E
for i in range(max(a.warmup, 60)):
    step()
    if i + 1 >= a.warmup and not E.tuning_pending():
        break
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
gf = 16 * 61.232 + 32.829
print(json.dumps({'metric': 'anonymised clips/s (%s fa + I3Res50, 16x224^2)' % a.arch_fa, 'value': round(a.clips / dt, 1), 'batch': a.batch,
                  **({'algorithmic_TFLOPs': round(a.clips * gf / dt / 1e3, 1), 'GFLOP_per_clip': round(gf, 1)} if a.arch_fa == 'unet' else {})}))
