"""Weight-gradient kernel timing on the UNet's 3 x 3 layer shapes at cfg3 (384 frames of 112^2). TEDSPAD_WGRAD_NO_ROWS=1 selects the
gather kernel for the 3 x 3 layers. Usage: python scripts/wgrad_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ted_spad_amd import engine as E, train_engine as TE
from ted_spad_amd.synth import synth_tensor

TE.WGRAD_STREAM = False
SHAPES = [("inc.1 / up4.1", 64, 64, 112), ("up4.0", 128, 64, 112), ("down1.1", 128, 128, 56), ("up3.0", 256, 128, 56), ("down2.1", 256, 256, 28),
          ("up2.0", 512, 256, 28), ("down3.1", 512, 512, 14), ("up1.0", 1024, 512, 14), ("down4.1", 512, 512, 7)]


def timed(fn, reps=7):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


n = 384
for name, cin, cout, hw in SHAPES:
    w = torch.nn.Parameter((synth_tensor(1, "w", (cout, cin, 3, 3), -1, 1) * 0.05).cuda())
    L = TE.ConvLayer(w, None, (1, 1, 1), (0, 1, 1))
    x = E.Act(torch.randn((n, 1, hw, hw, cin), device="cuda").to(torch.float16), cin)
    dy = E.Act(torch.randn((n, 1, hw, hw, cout), device="cuda").to(torch.float16), cout)
    TE.ARENA.reset("cuda")
    us = timed(lambda: L.wgrad(x, dy))
    fl = 2.0 * n * hw * hw * cout * cin * 9
    print("%-14s %4d -> %3d @ %3d^2: %7.0f us  %6.0f TFLOP/s" % (name, cin, cout, hw, us, fl / us / 1e6))
    del x, dy
