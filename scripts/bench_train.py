"""cfg3 (BASELINE.json configs[2]): one anonymizer training iteration per phase on 1 MI355X:
UNet anonymizer + I3Res50 + CE/triplet, batch 8 x 48 frames x 112^2, f16 activations.
Prints one JSON line with ms/step of both phases and the algorithmic TFLOP/s."""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ted_spad_amd.model_loaders import load_fa_model, load_ft_model
from ted_spad_amd.synth import synth_state_dict, synth_train_video
from ted_spad_amd.train_step import AnonymizerTrainStep

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--res', type=int, default=112)
ap.add_argument('--steps', type=int, default=6)
ap.add_argument('--warmup', type=int, default=30)   # the tile tuner needs ~30 calls per conv geometry
a = ap.parse_args()
import contextlib, io
with contextlib.redirect_stdout(io.StringIO()):
    fa, ft = load_fa_model(arch='unet'), load_ft_model('largei3d', num_classes=102)
fa.load_state_dict(synth_state_dict(fa.state_dict(), 0)); ft.load_state_dict(synth_state_dict(ft.state_dict(), 0))
fa, ft = fa.cuda(), ft.cuda()
step = AnonymizerTrainStep(fa, ft)
video = synth_train_video(0, 'bench_train', (a.batch, 48, 3, a.res, a.res), device='cuda')
labels = torch.randint(1, 102, (a.batch,), device='cuda')
res = {}
for name, fn in (('phase1_update_fa', step.step_fa), ('phase2_update_ft', step.step_ft)):
    for _ in range(a.warmup):
        fn(video, labels)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.steps):
        out = fn(video, labels)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
    res[name + '_ms'] = round(dt * 1e3, 2)
    res[name + '_loss'] = out['loss_ft']
scale = (a.res / 112.0) ** 2 * a.batch / 8.0
res['phase1_TFLOPs'] = round(18.040 * scale / (res['phase1_update_fa_ms'] / 1e3), 1)
res['phase2_TFLOPs'] = round(6.480 * scale / (res['phase2_update_ft_ms'] / 1e3), 1)
res['config'] = 'cfg3: UNet+I3Res50+CE/triplet, batch %dx48x%d^2, f16, 1 MI355X, fb branch excluded' % (a.batch, a.res)
print(json.dumps(res))
