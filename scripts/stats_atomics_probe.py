"""What the batch-statistics atomics of a training-forward conv cost: the UNet's 64 -> 64 and 128 -> 128 layers (384 frames) with and without `stats`.
Usage: python scripts/stats_atomics_probe.py"""
import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ted_spad_amd import engine as E
for hw, c in ((112, 64), (56, 128), (28, 256)):
    x = E.Act((torch.rand(384, 1, hw, hw, c, device='cuda') - 0.5).half(), c)
    pc = E.PackedConv((torch.rand(c, c, 1, 3, 3) - 0.5) * 0.1, torch.ones(c), torch.zeros(c), device='cuda')
    stats = torch.zeros((2, pc.cpad), device='cuda')
    for cfg in (32, 38):
        for st in (None, stats):
            E.FORCE_TILE_CFG = cfg
            try:
                out = pc(x, pads=(0, 1, 1), stats=st, relu=False); torch.cuda.synchronize()
            except Exception as e:
                print('cfg', cfg, 'n/a', str(e)[:60]); continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): pc(x, pads=(0, 1, 1), out=out, stats=st, relu=False)
            e1.record(); torch.cuda.synchronize()
            print('%3d^2 x %3d cfg %d stats %-5s: %.1f us' % (hw, c, cfg, st is not None, e0.elapsed_time(e1) * 100))
