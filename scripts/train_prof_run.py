"""The cfg3 training iteration (bench.py's train_cfg3: UNet fa + I3Res50 ft, batch 8 x 48 x 112^2, both phases alternating) for a rocprofv3 pass:
tunes, then brackets exactly K iterations with two launches of the clock-probe kernel (1 workgroup, 1 MFMA) that scripts/summarize_train.py uses as markers.
Usage: rocprofv3 ... -- python3 scripts/train_prof_run.py [K] [fb]      (fb: the whole train_epoch body -- privacy branch + NT-Xent on the VISPR views 2 x (12,3,224,224))"""
import os, sys, io, contextlib, ctypes as C, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ted_spad_amd import _lib, engine as E
from ted_spad_amd.model_loaders import load_fa_model, load_fb_model, load_ft_model
from ted_spad_amd.synth import synth_state_dict, synth_tensor, synth_train_video
from ted_spad_amd.train_step import AnonymizerTrainStep
K = int(sys.argv[1]) if len(sys.argv) > 1 else 3
FB = len(sys.argv) > 2 and sys.argv[2] == 'fb'
with contextlib.redirect_stdout(io.StringIO()):
    fa, ft = load_fa_model(arch='unet'), load_ft_model('largei3d', num_classes=102)
fa.load_state_dict(synth_state_dict(fa.state_dict(), 0)); ft.load_state_dict(synth_state_dict(ft.state_dict(), 0))
fa, ft = fa.cuda(), ft.cuda()
fb, views = None, None
if FB:
    with contextlib.redirect_stdout(io.StringIO()):
        fb = load_fb_model(arch='r50', ssl=True, pretrained=False)
    fb.load_state_dict(synth_state_dict(fb.state_dict(), 0)); fb = fb.cuda()
    views = [synth_tensor(0, 'vispr_view%d' % v, (12, 3, 224, 224), device='cuda') for v in range(2)]
_step = AnonymizerTrainStep(fa, ft, fb_model=fb)
class step:          # the two phases with the views bound
    step_fa = staticmethod(lambda v, l: _step.step_fa(v, l, views))
    step_ft = staticmethod(lambda v, l: _step.step_ft(v, l, inputs_vispr=views))
video = synth_train_video(0, 'bench_train', (8, 48, 3, 112, 112), device='cuda'); labels = torch.randint(1, 102, (8,), device='cuda')
for fn in (step.step_fa, step.step_ft):            # as bench.py: each phase until the tile tuner has settled every conv geometry
    for i in range(180):
        if i >= 45 and not E.tuning_pending():
            break
        fn(video, labels)
for _ in range(3): step.step_fa(video, labels); step.step_ft(video, labels)
torch.cuda.synchronize()
tbuf = torch.zeros(4, dtype=torch.int64, device='cuda')
def marker():
    _lib.check(_lib.lib().tedspad_clock_probe(1, 1, tbuf.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'marker')
marker()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(K): step.step_fa(video, labels); step.step_ft(video, labels)
e1.record()
marker()
torch.cuda.synchronize()
print('%d iterations, %.2f ms per iteration under the profiler' % (K, e0.elapsed_time(e1) / K))
