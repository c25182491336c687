"""Cycle stamps of the ping-pong conv kernel (debug build: hipcc -DTEDSPAD_P8_ABLATIONS -> libtedspad_hip_dbg.so): per workgroup
the prologue, the K loop, the two epilogue passes and the drain of the stores (s_memtime, 100 MHz-independent shader clock)."""
import os, sys, subprocess, ctypes as C, numpy as np, torch
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ted_spad_amd import _lib, build as B
# the stamped build of conv_p8.hip, linked with the release objects of everything else (built here, on the GPU box)
objs = [os.path.join(B.CSRC, 'build', os.path.basename(s)[:-4] + '.o') for s in B.sources()]
dbg_so, dbg_o = os.path.join(ROOT, 'ted_spad_amd', 'libtedspad_hip_dbg.so'), '/tmp/dbg_conv_p8.o'
subprocess.run([B.HIPCC] + B.FLAGS + ['-DTEDSPAD_P8_ABLATIONS', '-c', os.path.join(B.CSRC, 'conv_p8.hip'), '-o', dbg_o], check=True)
subprocess.run([B.HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', dbg_so] + [x if not x.endswith('conv_p8.o') else dbg_o for x in objs], check=True)
_lib.LIB_PATH = dbg_so
from ted_spad_amd import engine as E
L = _lib.lib()
L.tedspad_debug_set_p8_ts.restype = C.c_int32; L.tedspad_debug_set_p8_ts.argtypes = [C.c_void_p]
dbg = torch.zeros(65536 * 6, dtype=torch.int64, device='cuda')
assert L.tedspad_debug_set_p8_ts(dbg.data_ptr()) == 0
def probe(dims, cin, cout, k, pads, res=False):
    n, t, h, w = dims
    x = E.Act((torch.rand(n, t, h, w, cin, device='cuda') - 0.5).half(), cin)
    pc = E.PackedConv((torch.rand(cout, cin, *k) - 0.5) * 0.05, torch.ones(cout), torch.zeros(cout), device='cuda')
    r = E.Act((torch.rand(n, t, h, w, cout, device='cuda') - 0.5).half(), cout) if res else None
    E.FORCE_TILE_CFG = 25
    for _ in range(3): out = pc(x, pads=pads, residual=r)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dbg.zero_(); e0.record(); pc(x, pads=pads, out=out, residual=r); e1.record(); torch.cuda.synchronize()
    ts = dbg.cpu().numpy().reshape(-1, 6).astype(np.float64); ts = ts[ts[:, 0] > 0]
    d = np.diff(ts, axis=1)
    first = ts[:, 0] < np.percentile(ts[:, 0], 50)     # workgroups of the first round
    nk = k[0] * k[1] * k[2] * cin // 64
    print('M=%d N=%d K=%d: %.1f us, %d WGs | cycles (median) prologue %.0f  loop %.0f = %.0f / K tile  pass0 %.0f  pass1 %.0f  store drain %.0f | whole WG %.0f; kernel span %.0f cycles' % (
        n*t*h*w, cout, nk*64, e0.elapsed_time(e1) * 1e3, len(ts), *[np.median(d[:, i]) for i in range(1)], np.median(d[:, 1]), np.median(d[:, 1]) / nk,
        np.median(d[:, 2]), np.median(d[:, 3]), np.median(d[:, 4]), np.median(ts[:, 5] - ts[:, 0]), ts[:, 5].max() - ts[:, 0].min()))
    E.FORCE_TILE_CFG = None
c = int(sys.argv[1]) if len(sys.argv) > 1 else 225
# the ping-pong launches of one I3Res50 forward at bench size, as plain convs of the same (M, N, K)
probe((c, 1, 56, 56), 512, 256, (1, 1, 1), (0, 0, 0))              # layer2.0 conv1 (two frames folded into K and N)
probe((c, 1, 28, 28), 1024, 256, (1, 1, 1), (0, 0, 0))             # layer2.2 conv1
probe((c, 1, 28, 28), 1024, 512, (1, 1, 1), (0, 0, 0))             # layer3.0 conv1
probe((c, 2, 14, 14), 768, 1024, (1, 1, 1), (0, 0, 0))             # layer3.0 conv3 + downsample (K = 256 + 512)
probe((c, 2, 14, 14), 1024, 512, (1, 1, 1), (0, 0, 0))             # layer4.0 conv1
probe((c, 2, 7, 7), 512, 512, (1, 3, 3), (0, 1, 1))                # layer4 conv2
probe((c, 2, 7, 7), 1536, 2048, (1, 1, 1), (0, 0, 0))              # layer4.0 conv3 + downsample
probe((c, 2, 7, 7), 512, 2048, (1, 1, 1), (0, 0, 0), True)         # layer4 conv3 + residual
probe((c, 2, 7, 7), 2048, 512, (1, 1, 1), (0, 0, 0))               # layer4.2 conv1
probe((c, 2, 14, 14), 256, 1024, (1, 1, 1), (0, 0, 0), True)       # layer3 conv3 + residual (unfused form)
