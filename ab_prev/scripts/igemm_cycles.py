"""Cycle stamps of the generic implicit-GEMM conv kernel (debug build only: hipcc -DTEDSPAD_DEBUG_TS -> libtedspad_hip_dbg.so):
per workgroup the prologue, the K loop, and the part of the loop that wave 0 spent in `s_waitcnt vmcnt` + `s_barrier`."""
import os, sys, ctypes as C, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ted_spad_amd import _lib
_lib.LIB_PATH = _lib.LIB_PATH.replace('libtedspad_hip.so', 'libtedspad_hip_dbg.so')
from ted_spad_amd import engine as E
L = _lib.lib()
L.tedspad_debug_set_igemm_ts.restype = C.c_int32; L.tedspad_debug_set_igemm_ts.argtypes = [C.c_void_p]
dbg = torch.zeros(262144 * 4, dtype=torch.int64, device='cuda')
assert L.tedspad_debug_set_igemm_ts(dbg.data_ptr()) == 0
def probe(dims, cin, cout, k, pads, cfgs):
    n, t, h, w = dims
    x = E.Act((torch.rand(n, t, h, w, cin, device='cuda') - 0.5).half(), cin)
    pc = E.PackedConv((torch.rand(cout, cin, *k) - 0.5) * 0.05, torch.ones(cout), torch.zeros(cout), device='cuda')
    nk = (k[0] * k[1] * k[2] * cin + 63) // 64
    for cfg in cfgs:
        E.FORCE_TILE_CFG = cfg
        dbg.zero_()
        try:
            for _ in range(3): pc(x, pads=pads)
        except _lib.TedSpadHipError:
            continue
        torch.cuda.synchronize()
        ts = dbg.cpu().numpy().reshape(-1, 4).astype(np.float64); ts = ts[ts[:, 0] > 0]
        pro, loop, wait = ts[:, 1] - ts[:, 0], ts[:, 2] - ts[:, 1], ts[:, 3]
        print('M=%d N=%d K=%d cfg %2d: WGs %5d  prologue %6.0f  loop %6.0f = %4.0f/step  of which waiting (vmcnt+barrier, wave 0) %4.0f/step (%.0f %%)' % (
            n*t*h*w, cout, nk*64, cfg, len(ts), np.median(pro), np.median(loop), np.median(loop) / nk, np.median(wait) / max(nk - 1, 1), 100 * np.median(wait) / np.median(loop)))
    E.FORCE_TILE_CFG = None
probe((75, 2, 14, 14), 256, 256, (1, 3, 3), (0, 1, 1), [1, 3, 11, 13, 18, 24])
probe((75, 2, 28, 28), 128, 128, (1, 3, 3), (0, 1, 1), [1, 13, 17, 23])
probe((75, 4, 55, 55), 64, 64, (1, 3, 3), (0, 1, 1), [17, 7, 10])
