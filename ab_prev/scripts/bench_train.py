"""cfg3 (BASELINE.json configs[2]): one anonymizer training iteration per phase on 1 MI355X:
UNet anonymizer + I3Res50 + CE/triplet, batch 8 x 48 frames x 112^2, f16 activations; with --fb also the privacy
branch (ResNet-50 + MLP, NT-Xent) on two VISPR views of (12,3,224,224) -- the whole train_epoch body.
Prints one JSON line with ms/step of both phases and the algorithmic TFLOP/s."""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ted_spad_amd.model_loaders import load_fa_model, load_fb_model, load_ft_model
from ted_spad_amd.synth import synth_state_dict, synth_tensor, synth_train_video
from ted_spad_amd.train_step import AnonymizerTrainStep

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--res', type=int, default=112)
ap.add_argument('--steps', type=int, default=6)
ap.add_argument('--warmup', type=int, default=45)   # the tile tuner needs up to ~40 calls per conv geometry (convs called once per step)
ap.add_argument('--fb', action='store_true', help='include the privacy branch (2 x (12,3,224,224) VISPR views)')
ap.add_argument('--vispr-batch', type=int, default=12)
ap.add_argument('--vispr-res', type=int, default=224)
a = ap.parse_args()
import contextlib, io
with contextlib.redirect_stdout(io.StringIO()):
    fa, ft = load_fa_model(arch='unet'), load_ft_model('largei3d', num_classes=102)
fa.load_state_dict(synth_state_dict(fa.state_dict(), 0)); ft.load_state_dict(synth_state_dict(ft.state_dict(), 0))
fa, ft = fa.cuda(), ft.cuda()
fb, views = None, None
if a.fb:
    with contextlib.redirect_stdout(io.StringIO()):
        fb = load_fb_model(arch='r50', ssl=True)
    fb.load_state_dict(synth_state_dict(fb.state_dict(), 0))
    fb = fb.cuda()
    gain = (torch.arange(1, a.vispr_batch + 1, device='cuda').float() / a.vispr_batch).view(-1, 1, 1, 1)
    views = [synth_tensor(0, 'vispr%d' % i, (a.vispr_batch, 3, a.vispr_res, a.vispr_res), device='cuda') * gain for i in range(2)]
step = AnonymizerTrainStep(fa, ft, fb_model=fb)
video = synth_train_video(0, 'bench_train', (a.batch, 48, 3, a.res, a.res), device='cuda')
labels = torch.randint(1, 102, (a.batch,), device='cuda')
res = {}
for name, fn in (('phase1_update_fa', lambda v, l: step.step_fa(v, l, views)), ('phase2_update_ft', lambda v, l: step.step_ft(v, l, inputs_vispr=views)),
                 ('action_step_frozen_bn', lambda v, l: step.step_action(v, l))):
    from ted_spad_amd import engine as _E
    for i in range(4 * a.warmup):            # at least `warmup` steps, then until the tile tuner has settled every conv geometry
        if i >= a.warmup and not _E.tuning_pending():
            break
        fn(video, labels)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.steps):
        out = fn(video, labels)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
    res[name + '_ms'] = round(dt * 1e3, 2)
    res[name + '_loss'] = out['loss_ft'] if 'loss_ft' in out else out['loss']
    if out.get('loss_fb') is not None:
        res[name + '_loss_fb'] = out['loss_fb']
scale = (a.res / 112.0) ** 2 * a.batch / 8.0
# fb terms (algorithmic, SURVEY.md §8d counting): UNet 61.232 GFLOP/frame @224^2, ResNet-50 8.18 GFLOP/image @224^2
vs = (a.vispr_res / 224.0) ** 2 * 2 * a.vispr_batch / 1e3 if a.fb else 0.0
p1 = 18.040 * scale + vs * (3 * 61.232 + 2 * 8.18)         # phase 1: fa fwd+dgrad+wgrad, frozen fb fwd+dgrad
p2 = 6.480 * scale + vs * (1 * 61.232 + 3 * 8.18)          # phase 2: fa fwd only, fb fwd+dgrad+wgrad
res['phase1_TFLOPs'] = round(p1 / (res['phase1_update_fa_ms'] / 1e3), 1)
res['phase2_TFLOPs'] = round(p2 / (res['phase2_update_ft_ms'] / 1e3), 1)
res['config'] = 'cfg3: UNet+I3Res50+CE/triplet, batch %dx48x%d^2, f16, 1 MI355X, %s' % (
    a.batch, a.res, 'fb branch (ResNet-50+MLP, NT-Xent) on 2x(%d,3,%d,%d) views included' % (a.vispr_batch, a.vispr_res, a.vispr_res) if a.fb else 'fb branch excluded')
print(json.dumps(res))
