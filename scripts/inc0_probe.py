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
for cfg in range(0, _lib.lib().tedspad_conv_num_tile_cfgs() + 1):
    E.FORCE_TILE_CFG = cfg
    try:
        print("cfg %2d: %6.0f us" % (cfg, timed(lambda: L.forward(x))))
    except Exception as e:
        print("cfg %2d: n/a (%s)" % (cfg, str(e)[60:140]))
