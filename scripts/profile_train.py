import os, sys, torch, contextlib, io
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ted_spad_amd.model_loaders import load_fa_model, load_ft_model
from ted_spad_amd.synth import synth_state_dict, synth_train_video
from ted_spad_amd.train_step import AnonymizerTrainStep
with contextlib.redirect_stdout(io.StringIO()):
    fa, ft = load_fa_model(arch='unet'), load_ft_model('largei3d', num_classes=102)
fa.load_state_dict(synth_state_dict(fa.state_dict(), 0)); ft.load_state_dict(synth_state_dict(ft.state_dict(), 0))
fa, ft = fa.cuda(), ft.cuda()
step = AnonymizerTrainStep(fa, ft)
video = synth_train_video(0, 'bench_train', (8, 48, 3, 112, 112), device='cuda'); labels = torch.randint(1, 102, (8,), device='cuda')
from ted_spad_amd import engine as E
for fn in (step.step_fa, step.step_ft):            # as bench.py: each phase until the tile tuner has settled every conv geometry
    for i in range(180):
        if i >= 45 and not E.tuning_pending():
            break
        fn(video, labels)
for _ in range(5): step.step_fa(video, labels); step.step_ft(video, labels)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
for name, fn in (('phase1', step.step_fa), ('phase2', step.step_ft)):
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        fn(video, labels); torch.cuda.synchronize()
    ev = prof.key_averages()
    rows = sorted(ev, key=lambda e: -e.device_time_total)[:14]
    tot = sum(e.device_time_total for e in ev)
    from torch.autograd import DeviceType
    kern = [e for e in ev if e.device_type == DeviceType.CUDA]
    print(name, 'GPU kernels per step: %d launches, %.2f ms device time' % (sum(e.count for e in kern), sum(e.device_time_total for e in kern) / 1e3))
    print('  -- by device time')
    for e in sorted(kern, key=lambda e: -e.device_time_total)[:40]: print('   %-90s n=%4d  %8.2f ms' % (e.key[:90], e.count, e.device_time_total/1e3))
    pat = os.environ.get('PROFILE_KERNEL')           # every launch of the kernels whose name contains this, in launch order
    if pat:
        evs = sorted((e for e in prof.events() if e.device_type == DeviceType.CUDA and pat in e.name), key=lambda e: e.time_range.start)
        print('  -- %s: %d launches (us): %s' % (pat, len(evs), ' '.join('%.0f' % e.device_time for e in evs)))
        if os.environ.get('PROFILE_NAMES'):
            import re
            for e in [e for e in evs if e.device_time >= float(os.environ.get('PROFILE_MIN_US', '0'))][:int(os.environ['PROFILE_NAMES'])]:
                print('     %6.0f us  %s' % (e.device_time, re.sub(r'tedspad::|\(anonymous namespace\)::|void ', '', e.name)[:110]))
    print('  -- by launch count')
    for e in sorted(kern, key=lambda e: -e.count)[:45]: print('   %-90s n=%4d  %8.2f ms' % (e.key[:90], e.count, e.device_time_total/1e3))
