"""Where a workgroup of layer1's fused tail (csrc/conv_bneck.hip, conv_bneck_tail_kernel) spends its cycles: the kernel built with -DTEDSPAD_BT_STAGE_STAMPS
into libtedspad_hip_bt.so by this script on the GPU box (s_memtime at the stage boundaries of every workgroup). Usage: python scripts/bneck_tail_cycles.py [clips]"""
import os, sys, subprocess, ctypes as C, numpy as np, torch
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ted_spad_amd import _lib, build as B
from ted_spad_amd import engine as E
from ted_spad_amd.synth import synth_tensor
objs = [os.path.join(B.CSRC, 'build', os.path.basename(s)[:-4] + '.o') for s in B.sources()]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 225
flags = sys.argv[2:]
dev = 'cuda'
so = os.path.join(ROOT, 'ted_spad_amd', 'libtedspad_hip_bt.so')
o = '/tmp/bt_conv_bneck.o'
subprocess.run([B.HIPCC] + B.FLAGS + ['-DTEDSPAD_BT_STAGE_STAMPS'] + flags + ['-c', os.path.join(B.CSRC, 'conv_bneck.hip'), '-o', o], check=True)
subprocess.run([B.HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', so] + [x if not x.endswith('conv_bneck.o') else o for x in objs], check=True)
_lib.LIB_PATH = so                      # the engine then drives the stamped build
L = _lib.lib()
L.tedspad_debug_set_bt_stage_ts.restype = C.c_int32; L.tedspad_debug_set_bt_stage_ts.argtypes = [C.c_void_p]
w2 = synth_tensor(1, "w2", (64, 64, 1, 3, 3), -0.05, 0.05); w3 = synth_tensor(1, "w3", (256, 64, 1, 1, 1), -0.1, 0.1)
one64, zero64, one256, zero256 = torch.ones(64), torch.zeros(64), torch.ones(256), torch.zeros(256)
c2 = E.PackedConv(w2, one64, zero64, dtype="f16", device=dev)
tp = E.BneckTail(c2, w3, one256, zero256)
x = E.Act(synth_tensor(1, "x", (n, 4, 56, 56, 64), -1, 1, device=dev).half(), 64)
res = E.Act(synth_tensor(1, "r", (n, 4, 56, 56, 256), -1, 1, device=dev).half(), 256)
out = E.Act.empty(n, 4, 56, 56, 256, torch.float16, dev)
tiles = n * 49
dbg = torch.zeros(tiles * 16, dtype=torch.int64, device=dev)
for _ in range(5): tp(x, residual=res, out=out)
torch.cuda.synchronize()
assert L.tedspad_debug_set_bt_stage_ts(dbg.data_ptr()) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); tp(x, residual=res, out=out); e1.record(); e1.synchronize()
assert L.tedspad_debug_set_bt_stage_ts(None) == 0
ts = dbg.cpu().numpy().reshape(tiles, 16).astype(np.float64)
d = np.diff(ts[:, :9], axis=1)
clk = (ts[:, 8] - ts[:, 0]) / (ts[:, 15] - ts[:, 14]) * 100.0
names = ['halo + first weights land', 'stage A: 9 taps', 'A -> B: bn2/relu pack, conv3 weights land', 'B step 0', 'B step 1', 'B step 2', 'B step 3', 'drain']
print('%d clips, %d tiles: launch %.0f us, clock %.0f MHz, workgroup %.0f cycles (median), %.0f (mean)' % (n, tiles, e0.elapsed_time(e1) * 1e3, np.median(clk),
                                                                                                     np.median(ts[:, 8] - ts[:, 0]), np.mean(ts[:, 8] - ts[:, 0])))
for i, nm in enumerate(names):
    print('  %-44s median %7.0f  mean %7.0f  p90 %7.0f cycles' % (nm, np.median(d[:, i]), np.mean(d[:, i]), np.percentile(d[:, i], 90)))
print('  prologue: start -> first DMA %.0f, -> halo issued %.0f, -> masks / accumulators ready %.0f, -> own loads landed %.0f, -> barrier passed %.0f (medians)' % tuple(
    np.median(ts[:, j] - ts[:, 0]) for j in (9, 10, 11, 12, 1)))
t0 = ts[:, 0].min()
print('  first start .. last end: %.0f cycles; workgroup starts: p50 %.0f p99 %.0f' % (ts[:, 8].max() - t0, np.median(ts[:, 0] - t0), np.percentile(ts[:, 0] - t0, 99)))
