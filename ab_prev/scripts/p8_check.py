"""Ping-pong tile (tile_cfg 25) against the generic tiles on the I3Res50 layer shapes it applies to: bit-identical
output, isolated time of both."""
import os, sys, argparse, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ted_spad_amd import _lib
if os.environ.get('TEDSPAD_DBG_LIB'):   # ablation build: hipcc -DTEDSPAD_P8_ABLATIONS -> libtedspad_hip_dbg.so (TEDSPAD_P8_ABL selects)
    _lib.LIB_PATH = _lib.LIB_PATH.replace('libtedspad_hip.so', 'libtedspad_hip_dbg.so')
from ted_spad_amd import engine as E
ap = argparse.ArgumentParser()
ap.add_argument('--clips', type=int, default=300)
ap.add_argument('--cfgs', default='25,26,1')
ap.add_argument('--reps', type=int, default=20)
ap.add_argument('--only', type=int, default=-1)
a = ap.parse_args()
cfgs = [int(c) for c in a.cfgs.split(',')]
n = a.clips
shapes = [  # dims (n,t,h,w), cin, cout, k, pads, stride, residual
    ((n, 2, 14, 14), 256, 256, (1, 3, 3), (0, 1, 1), (1, 1, 1), False),
    ((n, 2, 14, 14), 1024, 256, (3, 1, 1), (1, 0, 0), (1, 1, 1), False),
    ((n, 2, 14, 14), 1024, 256, (1, 1, 1), (0, 0, 0), (1, 1, 1), False),
    ((n, 2, 14, 14), 256, 1024, (1, 1, 1), (0, 0, 0), (1, 1, 1), True),
    ((n, 2, 28, 28), 128, 512, (1, 1, 1), (0, 0, 0), (1, 1, 1), True),
    ((n, 2, 28, 28), 512, 1024, (1, 1, 1), (0, 0, 0), (1, 2, 2), False),
    ((n, 2, 7, 7), 512, 512, (1, 3, 3), (0, 1, 1), (1, 1, 1), False),
    ((n, 2, 7, 7), 2048, 512, (3, 1, 1), (1, 0, 0), (1, 1, 1), False),
    ((n, 2, 7, 7), 512, 2048, (1, 1, 1), (0, 0, 0), (1, 1, 1), True),
    ((n, 2, 14, 14), 256, 512, (1, 3, 3), (0, 1, 1), (1, 2, 2), False),
]
torch.manual_seed(0)
for dims, cin, cout, k, pads, stride, res in (shapes if a.only < 0 else shapes[a.only:a.only + 1]):
    x = E.Act((torch.rand(*dims, cin, device='cuda') - 0.5).half(), cin)
    wt = (torch.rand(cout, cin, *k) - 0.5) * 0.05
    pc = E.PackedConv(wt, torch.rand(cout) + 0.5, torch.rand(cout) - 0.5, stride=stride, device='cuda')
    outs, line = {}, []
    for cfg in cfgs:
        E.FORCE_TILE_CFG = cfg
        try:
            out = pc(x, pads=pads)
        except _lib.TedSpadHipError as e:
            continue
        resid = E.Act((torch.rand(*out.dims, cout, device='cuda') - 0.5).half(), cout) if res else None
        if res:
            torch.manual_seed(1)
            resid = E.Act((torch.rand(*out.dims, cout, device='cuda') - 0.5).half(), cout)
        out = pc(x, pads=pads, residual=resid)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps): pc(x, pads=pads, out=out, residual=resid)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        M = out.dims[0] * out.dims[1] * out.dims[2] * out.dims[3]; K = k[0] * k[1] * k[2] * cin
        outs[cfg] = out.buf.clone()
        line.append('c%d %.0fus %.0fTF' % (cfg, ms * 1e3, 2.0 * M * cout * K / ms / 1e9))
    ref = [c for c in cfgs if c in outs and c not in (25, 26, 27)]
    same = 'n/a'
    if 25 in outs and ref:
        same = 'bit-identical' if torch.equal(outs[25], outs[ref[0]]) else 'DIFF max %.3e' % float((outs[25].float() - outs[ref[0]].float()).abs().max())
    if 26 in outs and ref:
        d = (outs[26].float() - outs[ref[0]].float()).abs(); same += '; c26 max diff %.2e (%.3f %% of elements differ)' % (float(d.max()), 100.0 * float((d > 0).float().mean()))
    print('M=%d N=%d K=%d k=%s s=%s: %s | p8 vs c%s: %s' % (M, cout, K, k, stride, '  '.join(line), ref[0] if ref else '-', same), flush=True)
