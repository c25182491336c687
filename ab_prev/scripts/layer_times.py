"""Per-launch timing of one batch forward (HIP events around every C-ABI launch):
shape, time, achieved TFLOP/s (algorithmic MACs x 2) and minimal HBM bytes / time."""
import os, sys, argparse
import torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ted_spad_amd import engine as E
from ted_spad_amd.model_loaders import load_ft_model, load_fa_model
from ted_spad_amd.synth import synth_state_dict, synth_clips

ap = argparse.ArgumentParser()
ap.add_argument('--arch', default='largei3d')
ap.add_argument('--batch', type=int, default=50)
ap.add_argument('--res', type=int, default=224)
ap.add_argument('--reps', type=int, default=3)
args = ap.parse_args()

recs = []
orig_call = E.PackedConv.__call__
def timed_call(self, x, *a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = orig_call(self, x, *a, **k); e1.record()
    n, t, h, w = out.dims
    M = n * t * h * w
    K = self.k[0] * self.k[1] * self.k[2] * self.cin
    res = k.get('residual') is not None
    byt = x.dims[0]*x.dims[1]*x.dims[2]*x.dims[3]*self.cin*2 + M*self.cout*2*(2 if res else 1) + self.cout*K*2
    recs.append(('conv k%s s%s c%d' % (self.k, self.stride, (lambda v: v if isinstance(v, int) else -1)(list(self._cfgs.values())[-1])), M, self.cout, K, 2.0*M*self.cout_real*K, byt, e0, e1))
    return out
E.PackedConv.__call__ = timed_call
orig_pool = E.maxpool
def timed_pool(x, *a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = orig_pool(x, *a, **k); e1.record()
    n, t, h, w = out.dims
    byt = (x.dims[0]*x.dims[1]*x.dims[2]*x.dims[3] + n*t*h*w) * x.c * 2
    recs.append(('maxpool', n*t*h*w, x.c, 0, 0.0, byt, e0, e1))
    return out
E.maxpool = timed_pool
import ted_spad_amd.i3res50, ted_spad_amd.inception_i3d, ted_spad_amd.unet

if args.arch == 'unet':
    m = load_fa_model(arch='unet')
    x = torch.rand(args.batch, 3, args.res, args.res, device='cuda')
    fwd = lambda: m(x)
else:
    m = load_ft_model(args.arch, num_classes=102)
    x = synth_clips(0, args.batch, (3, 16, args.res, args.res), device='cuda')
    fwd = lambda: (m.extract_features if hasattr(m, 'extract_features') else m.i3d.extract_features)(x)
m.load_state_dict(synth_state_dict(m.state_dict(), 0)); m = m.cuda().eval()
with torch.no_grad():
    for _ in range(60): fwd()
    torch.cuda.synchronize()
    for _ in range(args.reps):
        recs.clear(); fwd(); torch.cuda.synchronize()
tot = 0; totf = 0
for name, M, N, K, fl, byt, e0, e1 in recs:
    ms = e0.elapsed_time(e1); tot += ms; totf += fl
    print('%-28s M=%8d N=%5d K=%5d  %8.1f us  %7.1f TF/s  %6.2f TB/s(min bytes)' % (name, M, N, K, ms*1e3, fl/ms/1e9, byt/ms/1e9))
print('sum %.3f ms for %d clips -> %.1f us/clip, %.1f TF/s' % (tot, args.batch, tot*1e3/args.batch, totf/tot/1e9))
