"""The anonymised extraction of bench.py's `anon_extract` (fa = unet++ on 16 frames of 224 x 224 per clip -> Q1 feed -> I3Res50.extract_features; 25 clips per
forward) for a rocprofv3 pass: tunes, then brackets K passes over N clips with two launches of the clock-probe kernel that scripts/summarize_train.py uses as markers.
Usage: rocprofv3 ... -- python3 scripts/anon_prof_run.py [K] [N]"""
import os, sys, io, contextlib, ctypes as C, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ted_spad_amd import _lib, engine as E, extraction
from ted_spad_amd.model_loaders import load_fa_model, load_ft_model
from ted_spad_amd.synth import synth_clips, synth_state_dict
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N = int(sys.argv[2]) if len(sys.argv) > 2 else 75
B = 75
with contextlib.redirect_stdout(io.StringIO()):
    fa, ft = load_fa_model(), load_ft_model('largei3d', num_classes=102)
fa.load_state_dict(synth_state_dict(fa.state_dict(), 0)); ft.load_state_dict(synth_state_dict(ft.state_dict(), 0))
fa, ft = fa.cuda().eval(), ft.cuda().eval()
clips = torch.empty((N, 16, 3, 224, 224), dtype=torch.float32, device='cuda')
for i in range(0, N, 25):
    k = min(25, N - i)
    clips[i:i + k] = synth_clips(0, k, (3, 16, 224, 224), device='cuda', first=i).view(k, 16, 3, 224, 224)
out = torch.empty((N, 2048), dtype=torch.float32, device='cuda')
def step():
    for i in range(0, N, 75):
        out[i:i + 75] = ft.i3d.extract_features(extraction.feed(clips[i:i + 75], fa, 'reference', fa_batch=B)).flatten(1)
tbuf = torch.zeros(4, dtype=torch.int64, device='cuda')
def marker():
    _lib.check(_lib.lib().tedspad_clock_probe(1, 1, tbuf.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'marker')
with torch.no_grad():
    for i in range(40):
        step()
        if i >= 2 and not E.tuning_pending():
            break
    torch.cuda.synchronize()
    marker()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K):
        step()
    e1.record()
    marker()
    torch.cuda.synchronize()
print('%d passes over %d clips, %.2f ms per pass under the profiler (%.1f clips/s)' % (K, N, e0.elapsed_time(e1) / K, N * K / e0.elapsed_time(e1) * 1e3))
