"""Achieved HBM rate of the train-mode BatchNorm passes (backward reduction, forward apply, backward apply) on the cfg3 shapes (UNet levels at 384 frames, fb ResNet-50 at 24 images). Usage: python scripts/bn_probe.py"""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ted_spad_amd import _lib
L = _lib.lib()
S = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for name, n, hw, c in (("unet 112^2 x64", 384, 112, 64), ("unet 56^2 x128", 384, 56, 128), ("unet 28^2 x256", 384, 28, 256), ("unet 14^2 x512", 384, 14, 512),
                       ("fa views 224^2 x64", 24, 224, 64), ("fb 56^2 x256", 24, 56, 256), ("fb 56^2 x64", 24, 56, 64), ("fb 28^2 x512", 24, 28, 512), ("fb 7^2 x2048", 24, 7, 2048)):
    px = n * hw * hw
    dy = torch.randn((px, c), device='cuda').half(); y = torch.randn((px, c), device='cuda').half(); z = torch.randn((px, c), device='cuda').half()
    mean = torch.zeros(c, device='cuda'); inv = torch.ones(c, device='cuda'); gam = torch.ones(c, device='cuda'); bet = torch.zeros(c, device='cuda')
    sums = torch.zeros((2, c), device='cuda')
    for relu_y in (True, False):
        us = timed(lambda: _lib.check(L.tedspad_bn_bwd_reduce(dy.data_ptr(), y.data_ptr() if relu_y else None, z.data_ptr(), 0, mean.data_ptr(), inv.data_ptr(), gam.data_ptr(), bet.data_ptr(),
                                                            sums.data_ptr(), c, px, c, c, c, c, 1, 1, 0, S), 'r'))
        by = px * c * 2 * (3 if relu_y else 2)
        print("%-20s bn_bwd_reduce (%s): %7.1f us  %5.2f TB/s" % (name, "dy,y,z" if relu_y else "dy,z; mask from z", us, by / us / 1e6))
    stats = torch.stack([z.float().sum(0), (z.float() ** 2).sum(0)]).contiguous()
    rm = torch.zeros(c, device='cuda'); rv = torch.ones(c, device='cuda'); mo = torch.zeros(c, device='cuda'); io = torch.zeros(c, device='cuda')
    yo = torch.empty_like(z)
    us = timed(lambda: _lib.check(L.tedspad_bn_train_apply(z.data_ptr(), 0, stats.data_ptr(), c, px, gam.data_ptr(), bet.data_ptr(), C.c_float(1e-5), C.c_float(0.1), rm.data_ptr(), rv.data_ptr(),
                                                         mo.data_ptr(), io.data_ptr(), c, None, yo.data_ptr(), px, c, c, 0, c, 1, 1, 0, S), 'a'))
    print("%-20s bn_train_apply (z -> y): %7.1f us  %5.2f TB/s" % (name, us, px * c * 4 / us / 1e6))
    dz = torch.empty_like(z)
    us = timed(lambda: _lib.check(L.tedspad_bn_bwd_apply(dy.data_ptr(), None, z.data_ptr(), 0, mean.data_ptr(), inv.data_ptr(), gam.data_ptr(), bet.data_ptr(), sums.data_ptr(), c, dz.data_ptr(), None,
                                                       None, 1, px, c, c, c, c, c, 0, 1, 1, 0, S), 'b'))
    print("%-20s bn_bwd_apply (dy, z -> dz; mask from z): %7.1f us  %5.2f TB/s" % (name, us, px * c * 6 / us / 1e6))
