"""layer3 plain bottleneck at bench size (225 clips x 2 frames of 14 x 14, 1024 -> 256 -> 1024): the whole-block launch (engine.BneckFrame) against the three
launches it replaces, for the 1x1x1 and the 3x1x1 (folded two-frame) conv1. Usage: python scripts/bneck_frame_probe.py [clips]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ted_spad_amd import engine as E
from ted_spad_amd.synth import synth_tensor
n = int(sys.argv[1]) if len(sys.argv) > 1 else 225
dev = "cuda"
x = E.Act(synth_tensor(1, "x", (n, 2, 14, 14, 1024), -1, 1, device=dev).half(), 1024)
one = lambda c: torch.ones(c, device=dev)
zero = lambda c: torch.zeros(c, device=dev)


def timed(fn, reps=9):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


for kt in (1, 3):
    w1 = synth_tensor(1, "w1%d" % kt, (256, 1024, kt, 1, 1), -0.03, 0.03)
    w2 = synth_tensor(1, "w2", (256, 256, 1, 3, 3), -0.03, 0.03)
    w3 = synth_tensor(1, "w3", (1024, 256, 1, 1, 1), -0.06, 0.06)
    c1 = E.TPairConv(w1, one(256), zero(256), dtype="f16", device=dev) if kt == 3 else E.PackedConv(w1, one(256), zero(256), dtype="f16", device=dev)
    c2 = E.PackedConv(w2, one(256), zero(256), dtype="f16", device=dev)
    c3 = E.PackedConv(w3, one(1024), zero(1024), dtype="f16", device=dev)
    f1 = (lambda: c1(x)) if kt == 3 else (lambda: c1(x, pads=(0, 0, 0)))
    for _ in range(60):
        h1 = f1(); h2 = c2(h1, pads=(0, 1, 1)); c3(h2, residual=x, relu=True)
    t1, t2, t3 = timed(f1), timed(lambda: c2(h1, pads=(0, 1, 1))), timed(lambda: c3(h2, residual=x, relu=True))
    bf = E.BneckFrame(w1, one(256), zero(256), w2, one(256), zero(256), w3, one(1024), zero(1024), dtype="f16", device=dev)
    tf = timed(lambda: bf(x))
    gf = 2.0 * n * 2 * 196 * (256 * 1024 * (2 if kt == 3 else 1) + 256 * 2304 + 1024 * 256) / 1e9
    print("conv1 %dx1x1: unfused %.0f + %.0f + %.0f = %.0f us; whole block %.0f us (%.0f TFLOP/s, %.2f TB/s of x twice + y)" % (
        kt, t1, t2, t3, t1 + t2 + t3, tf, gf / tf * 1e-3 * 1e3, 3 * n * 2 * 196 * 2048 / tf / 1e6))
