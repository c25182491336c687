"""The UNet's first conv (3 -> 64, 3 x 3, 384 frames of 112^2) on every tile configuration that takes it, without / with the batch-statistics epilogue."""
import os, sys, torch
sys.path.insert(0, "/root/repo")
from ted_spad_amd import engine as E, train_engine as TE, _lib
from ted_spad_amd.synth import synth_tensor
w = torch.nn.Parameter((synth_tensor(1, "w", (64, 3, 3, 3), -1, 1) * 0.2).cuda())
L = TE.ConvLayer(w, None, (1, 1, 1), (0, 1, 1))
x = E.Act(torch.rand((384, 1, 112, 112, 8), device="cuda").to(torch.float16), 8)
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]
for cfg in (0, 2, 7, 9, 20, 21, 29, 30, 31):
    E.FORCE_TILE_CFG = cfg
    try:
        st = torch.zeros((2, L.fwd_conv().cpad), device="cuda")
        print("cfg %2d: %6.0f us, %6.0f us with batch statistics" % (cfg, timed(lambda: L.forward(x)), timed(lambda: L.forward(x, stats=st))))
    except Exception as e:
        print("cfg %2d: n/a (%s)" % (cfg, str(e)[60:140]))
