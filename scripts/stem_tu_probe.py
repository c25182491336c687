"""Stem in temporal-unfolded form (engine.StemTU: layout + kernel) against the pixel-pair form (clip_to_act + PackedConv)."""
import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ted_spad_amd import engine as E
n = int(sys.argv[1]) if len(sys.argv) > 1 else 75
x = torch.rand(n, 3, 16, 224, 224, device='cuda') - 0.5
w = (torch.rand(64, 3, 5, 7, 7) - 0.5) * 0.1
tu = E.StemTU(w, torch.ones(64), torch.zeros(64), device='cuda')
pc = E.PackedConv(w, torch.ones(64), torch.zeros(64), stride=(2, 2, 2), device='cuda', pair_w=3)
def old():
    a = E.clip_to_act(x, cpad=4)
    return pc(a, pads=(2, 3, pc.pair_pw), pads_back=(2, 3, 1))
def timeit(f, reps=10):
    for _ in range(40): f()     # covers the tile tuner of the old form
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
t_old, t_new = timeit(old), timeit(lambda: tu(x))
xtu = tu.layout(x); a_old = E.clip_to_act(x, cpad=4)
print('layout: pair %.0f us, tu %.0f us; stem kernel: pair %.0f us, tu %.0f us' % (timeit(lambda: E.clip_to_act(x, cpad=4)), timeit(lambda: tu.layout(x)),
      timeit(lambda: pc(a_old, pads=(2, 3, pc.pair_pw), pads_back=(2, 3, 1))), timeit(lambda: tu.conv(xtu))))
print('%d clips: pixel-pair form (layout + stem) %.0f us = %.2f us/clip; temporal-unfolded %.0f us = %.2f us/clip' % (n, t_old, t_old / n, t_new, t_new / n))
