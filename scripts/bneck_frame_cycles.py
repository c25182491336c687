"""Ablations of the whole-bottleneck kernel (csrc/conv_bneck_frame.hip built with -DTEDSPAD_BF_ABLATE into libtedspad_hip_abl.so by this script on the GPU
box): the time of one launch at bench size with parts of the kernel switched off. (The -DTEDSPAD_BF_STAMPS build with per-phase cycle stamps cost a third
of the kernel's time and spilled; its numbers were read as proportions only and the script no longer runs it.)"""
import os, sys, subprocess, ctypes as C, numpy as np, torch
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ted_spad_amd import _lib, build as B
objs = []
for src in B.sources():
    o = os.path.join(B.CSRC, 'build', os.path.basename(src)[:-4] + '.o')
    assert os.path.exists(o), 'run ted_spad_amd/build.py first'
    objs.append(o)
from ted_spad_amd import engine as E
from ted_spad_amd.synth import synth_tensor
n = int(sys.argv[1]) if len(sys.argv) > 1 else 225
dev = 'cuda'
x = E.Act(synth_tensor(1, 'x', (n, 2, 14, 14, 1024), -1, 1, device=dev).half(), 1024)
one = lambda c: torch.ones(c, device=dev)
zero = lambda c: torch.zeros(c, device=dev)
# ---- ablations in a build WITHOUT the stamps (they cost a third of the time) ----
abl_so = os.path.join(ROOT, 'ted_spad_amd', 'libtedspad_hip_abl.so')
o = '/tmp/abl_conv_bneck_frame.o'
subprocess.run([B.HIPCC] + B.FLAGS + ['-DTEDSPAD_BF_ABLATE', '-c', os.path.join(B.CSRC, 'conv_bneck_frame.hip'), '-o', o], check=True)
subprocess.run([B.HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', abl_so] + [x if 'conv_bneck_frame' not in x else o for x in objs], check=True)
LA = C.CDLL(abl_so)
LA.tedspad_debug_set_bf_ablate.restype = C.c_int32; LA.tedspad_debug_set_bf_ablate.argtypes = [C.c_int32]
fwd = LA.tedspad_bneck_frame_fwd
fwd.restype = C.c_int32
fwd.argtypes = _lib.SYMBOLS['tedspad_bneck_frame_fwd'][1]
out = torch.empty_like(x.buf)


def launch(bf):
    rc = fwd(x.ptr, x.ld, out.data_ptr(), x.ld, n, 2, 14, 14, 1024, 256, bf.w1[0].data_ptr(), bf.w1[-1].data_ptr(), bf.steps1, bf.w23.data_ptr(),
             *[v.data_ptr() for v in bf.bn], 1, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0


def timed(fn, reps=9):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


for kt in (1,):
    w1 = synth_tensor(1, 'w1%d' % kt, (256, 1024, kt, 1, 1), -0.03, 0.03); w2 = synth_tensor(1, 'w2', (256, 256, 1, 3, 3), -0.03, 0.03); w3 = synth_tensor(1, 'w3', (1024, 256, 1, 1, 1), -0.06, 0.06)
    bf = E.BneckFrame(w1, one(256), zero(256), w2, one(256), zero(256), w3, one(1024), zero(1024), dtype='f16', device=dev)
    for bits, what in ((0, 'full'), (1, 'no weight DMA'), (3, 'no DMA at all'), (12, 'no residual loads, no stores'), (32, 'no epilogue arithmetic'),
                       (44, 'no residual, stores, epilogue arithmetic'), (47, 'MFMAs + LDS reads + barriers only'), (16, 'no MFMAs'), (63, 'skeleton: LDS reads + barriers'),
                       (63 + 64, 'skeleton without barriers: LDS reads + loop code'), (63 + 128, 'skeleton without LDS reads: barriers + loop code'),
                       (255, 'loop code only'), (47 + 64, 'MFMAs + LDS reads, no barriers'), (47 + 128, 'MFMAs + barriers, no LDS reads'), (64, 'everything but barriers')):
        assert LA.tedspad_debug_set_bf_ablate(bits) == 0
        print('conv1 %dx1x1 ablation %3d  %-50s %6.0f us' % (kt, bits, what, timed(lambda: launch(bf))))
LA.tedspad_debug_set_bf_ablate(0)
