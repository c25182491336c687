"""Cycle stamps of the halo-direct conv kernel (debug build only: hipcc -DTEDSPAD_DEBUG_TS -> libtedspad_hip_dbg.so)."""
import os, sys, ctypes as C, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ted_spad_amd import _lib
_lib.LIB_PATH = _lib.LIB_PATH.replace('libtedspad_hip.so', 'libtedspad_hip_dbg.so')
from ted_spad_amd import engine as E
L = _lib.lib()
L.tedspad_debug_set_halo_ts.restype = C.c_int32; L.tedspad_debug_set_halo_ts.argtypes = [C.c_void_p]
dbg = torch.zeros(65536 * 4, dtype=torch.int64, device='cuda')
assert L.tedspad_debug_set_halo_ts(dbg.data_ptr()) == 0
def probe(dims, cin, cout, k, pads, cfg):
    n, t, h, w = dims
    x = E.Act((torch.rand(n, t, h, w, cin, device='cuda') - 0.5).half(), cin)
    pc = E.PackedConv((torch.rand(cout, cin, *k) - 0.5) * 0.05, torch.ones(cout), torch.zeros(cout), device='cuda')
    E.FORCE_TILE_CFG = cfg
    dbg.zero_()
    for _ in range(3): pc(x, pads=pads)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): pc(x, pads=pads)
    e1.record(); torch.cuda.synchronize()
    ts = dbg.cpu().numpy().reshape(-1, 4).astype(np.float64); ts = ts[ts[:, 0] > 0]
    pro, loop = ts[:, 1] - ts[:, 0], ts[:, 2] - ts[:, 1]
    print('halo cfg %d M=%d N=%d K=%d: %.1f us; WGs %d; median cycles: prologue %.0f, loop %.0f = %.0f per 64-deep step (%d steps)' % (
        cfg, n*t*h*w, cout, k[0]*k[1]*k[2]*cin, e0.elapsed_time(e1) * 100, len(ts), np.median(pro), np.median(loop), np.median(loop) / ts[0, 3], ts[0, 3]))
probe((75, 2, 14, 14), 256, 256, (1, 3, 3), (0, 1, 1), 15)
probe((75, 2, 28, 28), 128, 128, (1, 3, 3), (0, 1, 1), 15)
probe((75, 4, 55, 55), 64, 64, (1, 3, 3), (0, 1, 1), 16)
probe((75, 2, 14, 14), 1024, 256, (3, 1, 1), (1, 0, 0), 15)
