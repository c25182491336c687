"""layer1 bottleneck tail at bench size (225 clips: 4 x 55 x 55 pixels each): the fused launch (engine.BneckTail, both store variants)
against the two launches it replaces (conv2 + conv3 / dual conv3). Usage: python scripts/bneck_probe.py [clips]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ted_spad_amd import engine as E
from ted_spad_amd.synth import synth_tensor
n = int(sys.argv[1]) if len(sys.argv) > 1 else 225
dev = "cuda"
x = E.Act(synth_tensor(1, "x", (n, 4, 55, 55, 64), -1, 1, device=dev).half(), 64)
x2 = E.Act(synth_tensor(1, "x2", (n, 4, 55, 55, 64), -1, 1, device=dev).half(), 64)
res = E.Act(synth_tensor(1, "r", (n, 4, 55, 55, 256), -1, 1, device=dev).half(), 256)
w2 = synth_tensor(1, "w2", (64, 64, 1, 3, 3), -0.05, 0.05); w3 = synth_tensor(1, "w3", (256, 64, 1, 1, 1), -0.1, 0.1); wd = synth_tensor(1, "wd", (256, 64, 1, 1, 1), -0.1, 0.1)
one64, zero64, one256, zero256 = torch.ones(64), torch.zeros(64), torch.ones(256), torch.zeros(256)
c2 = E.PackedConv(w2, one64, zero64, dtype="f16", device=dev); c3 = E.PackedConv(w3, one256, zero256, dtype="f16", device=dev); cd = E.PackedConv(wd, one256, zero256, dtype="f16", device=dev)
tp, td = E.BneckTail(c2, w3, one256, zero256), E.BneckTail(c2, w3, one256, zero256, wd, one256, zero256)

def timed(fn, reps=7):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]

for _ in range(60):
    h = c2(x, pads=(0, 1, 1)); c3(h, residual=res, relu=True)
print("unfused plain: conv2 %.0f us + conv3+res %.0f us" % (timed(lambda: c2(x, pads=(0, 1, 1))), timed(lambda: c3(h, residual=res, relu=True))))
print("unfused dual : conv2 (same) + dual pointwise %.0f us" % timed(lambda: c3.call_dual(h, cd, x2, relu=True)))
for v in (0, 1, 2, 3):
    E.BneckTail.VARIANT = v
    print("variant %d: fused plain %.0f us, fused dual %.0f us" % (v, timed(lambda: tp(x, residual=res)), timed(lambda: td(x, x2=x2))))
E.BneckTail.VARIANT = 3
print("maxpool2 fused: conv2 (above) + conv3+res+pool %.0f us unfused; fused tail + pool %.0f us" % (
    timed(lambda: c3.call_pool_t2(h, residual=res, relu=True)), timed(lambda: tp(x, residual=res, pool_t2=True))))
# ---- layer2's plain block: 128 mid channels at 28 x 28 x 2 frames ----
x128 = E.Act(synth_tensor(1, "x128", (n, 2, 28, 28, 128), -1, 1, device=dev).half(), 128)
res512 = E.Act(synth_tensor(1, "r512", (n, 2, 28, 28, 512), -1, 1, device=dev).half(), 512)
w2b = synth_tensor(1, "w2b", (128, 128, 1, 3, 3), -0.04, 0.04); w3b = synth_tensor(1, "w3b", (512, 128, 1, 1, 1), -0.08, 0.08)
one128, zero128, one512, zero512 = torch.ones(128), torch.zeros(128), torch.ones(512), torch.zeros(512)
c2b = E.PackedConv(w2b, one128, zero128, dtype="f16", device=dev); c3b = E.PackedConv(w3b, one512, zero512, dtype="f16", device=dev)
tb = E.BneckTail(c2b, w3b, one512, zero512)
for _ in range(60):
    hb = c2b(x128, pads=(0, 1, 1)); c3b(hb, residual=res512, relu=True)
print("layer2 block: conv2 %.0f us + conv3+res %.0f us unfused; fused %.0f us" % (
    timed(lambda: c2b(x128, pads=(0, 1, 1))), timed(lambda: c3b(hb, residual=res512, relu=True)), timed(lambda: tb(x128, residual=res512))))
