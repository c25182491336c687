"""Isolated timing of the fused unet++ tail (tedspad_unetpp_tail_fwd: x_0_3 + segmentation head) on 1200 frames of 224 x 224 (a 75-clip anonymizer forward)."""
import os, sys, io, contextlib, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ted_spad_amd import engine as E
from ted_spad_amd.model_loaders import load_fa_model
from ted_spad_amd.synth import synth_state_dict
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
with contextlib.redirect_stdout(io.StringIO()):
    fa = load_fa_model()
fa.load_state_dict(synth_state_dict(fa.state_dict(), 0)); fa = fa.cuda().eval()
P = fa.packed()
x02 = E.Act((torch.rand(n, 1, 112, 112, 64, device='cuda') - 0.3).half(), 64)
with torch.no_grad():
    for _ in range(3): y = fa._tail(x02, P)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): y = fa._tail(x02, P)
    e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print('tail %d frames: %.3f ms  %.1f TFLOP/s algorithmic  checksum %.6f' % (n, ms, n * 2.86 / ms, float(y.double().sum())))
