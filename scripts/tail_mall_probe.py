"""Does the 256 MiB Infinity Cache serve layer1's tail faster than HBM? The fused plain tail (engine.BneckTail) timed back to back on the SAME tensors at
clip counts whose working set (mid 1.6 + residual 6.4 + output 6.4 MB per clip) does / does not fit. Usage: python scripts/tail_mall_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ted_spad_amd import engine as E
from ted_spad_amd.synth import synth_tensor
dev = "cuda"
w2 = synth_tensor(1, "w2", (64, 64, 1, 3, 3), -0.05, 0.05); w3 = synth_tensor(1, "w3", (256, 64, 1, 1, 1), -0.1, 0.1)
one64, zero64, one256, zero256 = torch.ones(64), torch.zeros(64), torch.ones(256), torch.zeros(256)
c2 = E.PackedConv(w2, one64, zero64, dtype="f16", device=dev)
tp = E.BneckTail(c2, w3, one256, zero256)
for n in (5, 10, 15, 21, 31, 52, 104, 225):
    x = E.Act(synth_tensor(1, "x", (n, 4, 56, 56, 64), -1, 1, device=dev).half(), 64)
    res = E.Act(synth_tensor(1, "r", (n, 4, 56, 56, 256), -1, 1, device=dev).half(), 256)
    out = E.Act.empty(n, 4, 56, 56, 256, torch.float16, dev)
    for _ in range(5): tp(x, residual=res, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = max(3, 900 // n)
    e0.record()
    for _ in range(reps): tp(x, residual=res, out=out)
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print("n=%3d: %.1f us per launch, %.2f us per clip, %.2f TB/s of minimum bytes (%d tiles, working set %d MB)" % (
        n, us, us / n, n * 12544 * 1152 / us / 1e6, n * 49, n * 14.4), flush=True)
