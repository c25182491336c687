"""Cycle stamps of the persistent ping-pong two-patch kernel (tile_cfg 40; stamp build: hipcc -DTEDSPAD_P3_STAMPS -> libtedspad_hip_dbg.so): per wave of workgroup 0, the
third tile's phases: LOAD start | reads + halo DMA issued | weights DMA + epilogue done | (lgkmcnt, barrier) COMPUTE start | MFMAs issued | DMA waited (then barrier)."""
import os, sys, subprocess, ctypes as C, numpy as np, torch
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ted_spad_amd import _lib, build as B
objs = [os.path.join(B.CSRC, 'build', os.path.basename(s)[:-4] + '.o') for s in B.sources()]
dbg_so, dbg_o = os.path.join(ROOT, 'ted_spad_amd', 'libtedspad_hip_dbg.so'), '/tmp/dbg_conv_patch3.o'
subprocess.run([B.HIPCC] + B.FLAGS + ['-DTEDSPAD_P3_STAMPS', '-c', os.path.join(B.CSRC, 'conv_patch3.hip'), '-o', dbg_o], check=True)
subprocess.run([B.HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', dbg_so] + [x if not x.endswith('conv_patch3.o') else dbg_o for x in objs], check=True)
_lib.LIB_PATH = dbg_so
from ted_spad_amd import engine as E
L = _lib.lib()
L.tedspad_debug_set_p3_ts.restype = C.c_int32; L.tedspad_debug_set_p3_ts.argtypes = [C.c_void_p]
dbg = torch.zeros(8 * 64, dtype=torch.int64, device='cuda')
assert L.tedspad_debug_set_p3_ts(dbg.data_ptr()) == 0
def probe(dims, cin, cout):
    n, t, h, w = dims
    x = E.Act((torch.rand(n, t, h, w, cin, device='cuda') - 0.5).half(), cin)
    pc = E.PackedConv((torch.rand(cout, cin, 1, 3, 3) - 0.5) * 0.05, torch.ones(cout), torch.zeros(cout), device='cuda')
    E.FORCE_TILE_CFG = 40
    for _ in range(3): out = pc(x, pads=(0, 1, 1))
    torch.cuda.synchronize()
    dbg.zero_(); pc(x, pads=(0, 1, 1), out=out); torch.cuda.synchronize()
    ts = dbg.cpu().numpy().reshape(8, 64).astype(np.float64)
    nph = min(cin // 32 * 3, 10)
    print('cin %d: per wave and phase of the tile: [load: reads+halo | wdma+epilogue | lgkm+barrier] [compute: mfma | wait | barrier->next load]' % cin)
    for wv in (0, 1, 4, 5):
        row = []
        for ph in range(nph):
            s = ts[wv, ph * 6:ph * 6 + 6]
            nxt = ts[wv, (ph + 1) * 6] if ph + 1 < nph and ts[wv, (ph + 1) * 6] > 0 else np.nan
            row.append('%4.0f %4.0f %4.0f | %4.0f %4.0f %4.0f' % (s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3], s[5] - s[4], nxt - s[5]))
        print('  wave %d: ' % wv + '  ||  '.join(row))
    E.FORCE_TILE_CFG = None
probe((400, 1, 112, 112), 64, 64)
probe((400, 1, 112, 112), 128, 64)
