"""Timing of the training step's gather-form backward passes at the UNet's largest shapes (cfg3: 384 frames): bilinear x2 upsample backward and 2 x 2 max-pool
backward. Prints a checksum of each result (an A/B of two library builds must print the same bits). Usage: python scripts/train_elemwise_probe.py"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ted_spad_amd import engine as E, train_engine as TE
from ted_spad_amd.synth import synth_tensor

def timed(fn, reps=10):
    for _ in range(2): out = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): out = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps, out

n = 384
for h, c in ((56, 64), (28, 128), (14, 256), (7, 512)):
    dy = E.Act(synth_tensor(3, "updy%d" % h, (n, 1, 2 * h, 2 * h, c), -1, 1, device="cuda").half(), c)
    us, dx = timed(lambda: TE.upsample2x_bwd(dy, h, h))
    gb = (dy.buf.numel() + dx.buf.numel()) * 2 / 1e9
    print("upsample2x_bwd %3d -> %3d, %3d ch: %7.1f us  %.2f TB/s  checksum %.6e" % (2 * h, h, c, us, gb / us * 1e3, float(dx.buf.double().sum())))
for h, c in ((112, 64), (56, 128), (28, 256), (14, 512)):
    x = E.Act(synth_tensor(3, "mpx%d" % h, (n, 1, h, h, c), 0, 1, device="cuda").half(), c)
    y, idx = E.maxpool(x, (1, 2, 2), (1, 2, 2), return_idx=True)
    dy = E.Act(synth_tensor(3, "mpdy%d" % h, (n, 1, h // 2, h // 2, c), -1, 1, device="cuda").half(), c)
    us, dx = timed(lambda: TE.maxpool_bwd(x, idx, dy, (1, 2, 2), (1, 2, 2)))
    gb = (dy.buf.numel() * 2 + idx.numel() * idx.element_size() + dx.buf.numel() * 2) / 1e9
    print("maxpool_bwd    %3d -> %3d, %3d ch: %7.1f us  %.2f TB/s  checksum %.6e" % (h // 2, h, c, us, gb / us * 1e3, float(dx.buf.double().sum())))
