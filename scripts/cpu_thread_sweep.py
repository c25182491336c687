import os, sys, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from oracle import i3res50_ref
from ted_spad_amd.model_loaders import wrapper_i3d
from ted_spad_amd.synth import synth_state_dict, synth_clips
sd = synth_state_dict(wrapper_i3d(102).state_dict(), 0)
sdc = {k[4:]: v for k, v in sd.items() if k.startswith("i3d.")}
x = synth_clips(0, 10)
print('cpus', os.cpu_count())
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    with torch.no_grad():
        i3res50_ref.extract_features(x[:2], sdc)
        t=time.perf_counter(); i3res50_ref.extract_features(x, sdc); dt=time.perf_counter()-t
    print(nt, 'threads', 10/dt, 'clips/s', flush=True)
