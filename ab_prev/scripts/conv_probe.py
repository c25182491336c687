"""Run ONE conv geometry repeatedly with a forced tile configuration (for rocprofv3 --pmc runs), or sweep all
tile configurations of that geometry in isolation (--sweep)."""
import os, sys, argparse, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ted_spad_amd import engine as E, _lib
ap = argparse.ArgumentParser()
ap.add_argument('--dims', default='100,2,14,14'); ap.add_argument('--cin', type=int, default=256); ap.add_argument('--cout', type=int, default=256)
ap.add_argument('--k', default='1,3,3'); ap.add_argument('--pads', default='0,1,1'); ap.add_argument('--cfg', type=int, default=1); ap.add_argument('--reps', type=int, default=20)
ap.add_argument('--sweep', action='store_true'); ap.add_argument('--res', action='store_true', help='fused residual input')
a = ap.parse_args()
n, t, h, w = map(int, a.dims.split(',')); k = tuple(map(int, a.k.split(','))); pads = tuple(map(int, a.pads.split(',')))
x = E.Act((torch.rand(n, t, h, w, a.cin, device='cuda') - 0.5).half(), a.cin)
wt = (torch.rand(a.cout, a.cin, *k) - 0.5) * 0.05
pc = E.PackedConv(wt, torch.ones(a.cout), torch.zeros(a.cout), device='cuda')
M = n*t*h*w; K = k[0]*k[1]*k[2]*a.cin
resid = E.Act((torch.rand(n, t, h, w, a.cout, device='cuda') - 0.5).half(), a.cout) if a.res else None
res = []
for cfg in (range(1, _lib.lib().tedspad_conv_num_tile_cfgs() + 1) if a.sweep else [a.cfg]):
    E.FORCE_TILE_CFG = cfg
    try:
        out = pc(x, pads=pads, residual=resid)
    except _lib.TedSpadHipError:
        continue
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps): pc(x, pads=pads, out=out, residual=resid)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    res.append((ms, cfg))
    if not a.sweep:
        print('cfg %d: %.1f us  %.1f TF/s' % (cfg, ms*1e3, 2.0*M*a.cout*K/ms/1e9))
if a.sweep:
    res.sort()
    print('M=%d N=%d K=%d k=%s: ' % (M, a.cout, K, k) + '  '.join('c%d %.0fus %.0fTF' % (c, ms*1e3, 2.0*M*a.cout*K/ms/1e9) for ms, c in res[:6]))
