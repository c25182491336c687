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

# ---- stage-boundary stamps (no measurable cost): cycles per stage and the clock; with the ablation switches compiled in as well ----
def stage_report(tag, flags, ablations):
    so = os.path.join(ROOT, 'ted_spad_amd', 'libtedspad_hip_%s.so' % tag)
    o = '/tmp/%s_conv_bneck_frame.o' % tag
    subprocess.run([B.HIPCC] + B.FLAGS + flags + ['-c', os.path.join(B.CSRC, 'conv_bneck_frame.hip'), '-o', o], check=True)
    subprocess.run([B.HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', so] + [x if 'conv_bneck_frame' not in x else o for x in objs], check=True)
    LS = C.CDLL(so)
    LS.tedspad_debug_set_bf_stage_ts.restype = C.c_int32; LS.tedspad_debug_set_bf_stage_ts.argtypes = [C.c_void_p]
    f = LS.tedspad_bneck_frame_fwd
    f.restype = C.c_int32
    f.argtypes = _lib.SYMBOLS['tedspad_bneck_frame_fwd'][1]
    dbg = torch.zeros(n * 2 * 2 * 8, dtype=torch.int64, device=dev)

    def go(bfk):
        assert f(x.ptr, x.ld, out.data_ptr(), x.ld, n, 2, 14, 14, 1024, 256, bfk.w1[0].data_ptr(), bfk.w1[-1].data_ptr(), bfk.steps1, bfk.w23.data_ptr(),
                 *[v.data_ptr() for v in bfk.bn], 1, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    for kt in (1, 3):
        w1 = synth_tensor(1, 'w1%d' % kt, (256, 1024, kt, 1, 1), -0.03, 0.03); w2 = synth_tensor(1, 'w2', (256, 256, 1, 3, 3), -0.03, 0.03); w3 = synth_tensor(1, 'w3', (1024, 256, 1, 1, 1), -0.06, 0.06)
        bfk = E.BneckFrame(w1, one(256), zero(256), w2, one(256), zero(256), w3, one(1024), zero(1024), dtype='f16', device=dev)
        for bits, what in ablations:
            if bits is not None and hasattr(LS, 'tedspad_debug_set_bf_ablate'):
                LS.tedspad_debug_set_bf_ablate.restype = C.c_int32; LS.tedspad_debug_set_bf_ablate.argtypes = [C.c_int32]
                assert LS.tedspad_debug_set_bf_ablate(bits) == 0
            for _ in range(20): go(bfk)
            torch.cuda.synchronize()
            assert LS.tedspad_debug_set_bf_stage_ts(dbg.data_ptr()) == 0
            dbg.zero_()
            us = timed(lambda: go(bfk), reps=5)
            ts = dbg.cpu().numpy().reshape(-1, 2, 8).astype(np.float64)
            assert LS.tedspad_debug_set_bf_stage_ts(None) == 0
            d = np.diff(ts[:, :, :5], axis=2)                         # stamps: start, end of stage 1, end of stage 2, start of stage 3, end
            clk = (ts[:, 0, 4] - ts[:, 0, 0]) / (ts[:, 0, 6] - ts[:, 0, 5]) * 100.0
            steps = [32 * (2 if kt == 3 else 1), 72, 32]
            print('%s conv1 %dx1x1 %-42s %4.0f us, %4.0f MHz, workgroup %6.0f cycles | per K step: stage 1 %5.0f  stage 2 %5.0f  stage 3 %5.0f | 2->3 transition %5.0f' % (
                tag, kt, what, us, np.median(clk), np.median(ts[:, 0, 4] - ts[:, 0, 0]), np.median(d[:, 0, 0]) / steps[0], np.median(d[:, 0, 1]) / steps[1],
                np.median(d[:, 0, 3]) / steps[2], np.median(d[:, 0, 2])))


stage_report('st', ['-DTEDSPAD_BF_STAGE_STAMPS'], [(None, 'release + stage stamps')])
for bits, what in ((4, 'no residual loads'), (8, 'no stores'), (12, 'no residual loads, no stores'), (32, 'no epilogue arithmetic'), (44, 'no residual, stores, epilogue arithmetic'),
                   (1, 'no weight DMA'), (2, 'no pixel DMA'), (16, 'no MFMAs')):
    stage_report('ct%d' % bits, ['-DTEDSPAD_BF_STAGE_STAMPS', '-DTEDSPAD_BF_CT_ABLATE=%d' % bits], [(None, 'compile-time ablation: ' + what)])
stage_report('sta', ['-DTEDSPAD_BF_STAGE_STAMPS', '-DTEDSPAD_BF_ABLATE'],
             [(0, 'ablate build, nothing off'), (32, 'no epilogue arithmetic'), (12, 'no residual loads, no stores'), (44, 'neither'), (1, 'no weight DMA'), (3, 'no DMA')])
# ---- how many pixel fragments of stages 1 / 2 the LOAD phase reads (the rest goes into the COMPUTE phase, three tiles ahead of its MFMAs) ----
for nb in (7, 5, 4, 3, 2):
    so = os.path.join(ROOT, 'ted_spad_amd', 'libtedspad_hip_nb%d.so' % nb)
    o = '/tmp/nb%d_conv_bneck_frame.o' % nb
    subprocess.run([B.HIPCC] + B.FLAGS + ['-DTEDSPAD_BF_NB12=%d' % nb, '-c', os.path.join(B.CSRC, 'conv_bneck_frame.hip'), '-o', o], check=True)
    subprocess.run([B.HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', so] + [x if 'conv_bneck_frame' not in x else o for x in objs], check=True)
    LN = C.CDLL(so)
    fwd = LN.tedspad_bneck_frame_fwd
    fwd.restype = C.c_int32
    fwd.argtypes = _lib.SYMBOLS['tedspad_bneck_frame_fwd'][1]
    print('release build, %d pixel fragments in LOAD (stages 1, 2): %6.0f us' % (nb, timed(lambda: launch(bf))))
