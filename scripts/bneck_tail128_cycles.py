"""Stage stamps of layer2's fused tail (conv_bneck_tail128_kernel), as scripts/bneck_tail_cycles.py does for layer1's. Usage: python scripts/bneck_tail128_cycles.py [clips] [-D...]"""
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
_lib.LIB_PATH = so
L = _lib.lib()
L.tedspad_debug_set_bt_stage_ts.restype = C.c_int32; L.tedspad_debug_set_bt_stage_ts.argtypes = [C.c_void_p]
w2 = synth_tensor(1, "w2b", (128, 128, 1, 3, 3), -0.04, 0.04); w3 = synth_tensor(1, "w3b", (512, 128, 1, 1, 1), -0.08, 0.08)
one128, zero128, one512, zero512 = torch.ones(128), torch.zeros(128), torch.ones(512), torch.zeros(512)
c2 = E.PackedConv(w2, one128, zero128, dtype="f16", device=dev)
tp = E.BneckTail(c2, w3, one512, zero512)
x = E.Act(synth_tensor(1, "x128", (n, 2, 28, 28, 128), -1, 1, device=dev).half(), 128)
res = E.Act(synth_tensor(1, "r512", (n, 2, 28, 28, 512), -1, 1, device=dev).half(), 512)
out = E.Act.empty(n, 2, 28, 28, 512, torch.float16, dev)
tiles = (n * 2 * 784 + 255) // 256
dbg = torch.zeros(tiles * 16, dtype=torch.int64, device=dev)
for _ in range(5): tp(x, residual=res, out=out)
torch.cuda.synchronize()
assert L.tedspad_debug_set_bt_stage_ts(dbg.data_ptr()) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); tp(x, residual=res, out=out); e1.record(); e1.synchronize()
assert L.tedspad_debug_set_bt_stage_ts(None) == 0
ts = dbg.cpu().numpy().reshape(tiles, 16).astype(np.float64)
clk = (ts[:, 8] - ts[:, 0]) / (ts[:, 15] - ts[:, 14]) * 100.0
seg = [('start -> first halo + weights landed (chunk 0)', 0, 1), ('chunk 0: 9 taps', 1, 9), ('chunk 1: halo lands', 9, 10), ('chunk 1: 9 taps', 10, 2),
       ('A -> B: bn2 / ReLU pack, conv3 weights land', 2, 3), ('B group 0', 3, 4), ('B group 1', 4, 5), ('B group 2 (with the weight reload)', 5, 6), ('B group 3', 6, 7), ('B groups 4..7', 7, 8)]
print('%d clips, %d tiles: launch %.0f us, clock %.0f MHz, workgroup %.0f cycles (median)' % (n, tiles, e0.elapsed_time(e1) * 1e3, np.median(clk), np.median(ts[:, 8] - ts[:, 0])))
for nm, a, b in seg:
    d = ts[:, b] - ts[:, a]
    print('  %-50s median %7.0f  p90 %7.0f cycles' % (nm, np.median(d), np.percentile(d, 90)))
