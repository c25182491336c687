"""Cost of the batch-statistics atomics in the conv epilogue (train-mode forward): one UNet-sized conv with and without `stats`.
Usage: python scripts/stats_probe.py [cin cout n hw]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ted_spad_amd import engine as E, train_engine as TE
from ted_spad_amd.synth import synth_tensor

cin, cout, n, hw = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (64, 64, 384, 112)
w = torch.nn.Parameter((synth_tensor(1, "w", (cout, cin, 3, 3), -1, 1) * 0.05).cuda())
L = TE.ConvLayer(w, None, (1, 1, 1), (0, 1, 1))
x = E.Act(synth_tensor(1, "x", (n, 1, hw, hw, cin), -1, 1).to(torch.float16).cuda(), cin)
pc = L.fwd_conv()


def timed(fn, reps=7):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


stats = torch.zeros((2, pc.cpad), device="cuda")
for _ in range(150):                       # tuner
    L.forward(x); L.forward(x, stats=stats)
for cfg in (None, 32, 33, 3, 0):
    E.FORCE_TILE_CFG = cfg
    try:
        a, b = timed(lambda: L.forward(x)), timed(lambda: L.forward(x, stats=stats))
    except Exception as e:
        print("cfg", cfg, "n/a", str(e)[:60]); continue
    print("cfg %s: %.0f us without stats, %.0f us with (%d x %d^2 x %d -> %d)" % (cfg, a, b, n, hw, cin, cout))
