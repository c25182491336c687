"""Timing of a plain layer1 block at bench size: temporal conv1 + fused tail (two launches) against the whole-block kernel (csrc/conv_bneck_l1.hip).
Usage: python scripts/bneck_l1_probe.py [clips]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ted_spad_amd import engine as E
from ted_spad_amd.synth import synth_tensor

n = int(sys.argv[1]) if len(sys.argv) > 1 else 225
t, h, w = 4, 55, 55
x = E.Act((torch.rand((n, t, h, w, 256), device="cuda") - 0.3).half(), 256)
w1 = synth_tensor(9, "w1", (64, 256, 3, 1, 1), -1, 1) * (2.0 / 768) ** 0.5
w2 = synth_tensor(9, "w2", (64, 64, 1, 3, 3), -1, 1) * (2.0 / 576) ** 0.5
w3 = synth_tensor(9, "w3", (256, 64, 1, 1, 1), -1, 1) * (2.0 / 64) ** 0.5
v = lambda k, c, lo, hi: synth_tensor(9, k, (c,), lo, hi)
s1, b1, s2, b2, s3, b3 = v("s1", 64, .5, 1.5), v("b1", 64, -.3, .3), v("s2", 64, .5, 1.5), v("b2", 64, -.3, .3), v("s3", 256, .5, 1.5), v("b3", 256, -.3, .3)
c1 = E.PackedConv(w1, s1, b1, dtype="f16", device="cuda")
c2 = E.PackedConv(w2, s2, b2, dtype="f16", device="cuda")
tail = E.BneckTail(c2, w3, s3, b3)
blk = E.BneckL1(w1, s1, b1, w2, s2, b2, w3, s3, b3, dtype="f16", device="cuda")


def timed(fn, reps=7):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


for _ in range(40):      # let the tuner settle
    m = c1(x, pads=(1, 0, 0))
old = tail(m, pads=(0, 1, 1), residual=x)
var = int(os.environ.get('L1_VARIANT', '1'))
new = blk(x, variant=var)
d = (old.buf.float() - new.buf.float()).abs()
print("max |diff| vs the two launches %.4g (max |value| %.3g)" % (float(d.max()), float(old.buf.float().abs().max())))
flop = 2.0 * n * t * h * w * (768 * 64 + 576 * 64 + 64 * 256)
t1, t2, t3 = timed(lambda: c1(x, pads=(1, 0, 0))), timed(lambda: tail(m, pads=(0, 1, 1), residual=x)), timed(lambda: blk(x, variant=var))
print("variant %d: conv1 %.0f us + tail %.0f us = %.0f us; whole block %.0f us (%.0f TFLOP/s algorithmic); pooled %.0f us" % (var, t1, t2, t1 + t2, t3, flop / t3 * 1e-6, timed(lambda: blk(x, pool_t2=True, variant=var))))
if os.environ.get("TEDSPAD_L1_ABLATE_SWEEP"):
    pass
# stage stamps of every workgroup (wave 0): cycles between the stage boundaries
nwg = n * (14 if var & 1 else 7) * 4
buf = torch.zeros((nwg, 8), dtype=torch.int64, device="cuda")
os.environ["TEDSPAD_L1_STAMPS"] = str(buf.data_ptr())
blk(x, variant=var); torch.cuda.synchronize()
del os.environ["TEDSPAD_L1_STAMPS"]
st = buf.cpu().double()
d = (st[:, 1:] - st[:, :-1]).mean(0)
names = ["first chunk lands", "stage 1 loop", "M1 write + barrier", "stage 2", "M2 write + barrier", "stage 3", "last stores drain"]
print("cycles per workgroup (mean over %d): " % nwg + ", ".join("%s %.0f" % (a, float(b)) for a, b in zip(names, d)) + "; total %.0f" % float((st[:, 7] - st[:, 0]).mean()))
