"""Timing of tedspad_frames_crop_resize_tp alone: 375 clips of 16 frames from 240 x 320 uint8 frames -> stem records. Usage: python scripts/crop_tp_probe.py"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ted_spad_amd import engine as E, preprocess
n = 375
frames = torch.randint(0, 256, (n * 32, 240, 320, 3), dtype=torch.uint8, device="cuda")
if os.environ.get("SMOOTH"):
    frames = (frames // 64) * 64
stem = E.StemPT(torch.zeros(64, 3, 5, 7, 7), torch.ones(64), torch.zeros(64), device="cuda")
box = preprocess.center_crop_box(240, 320, 192, 256)
rec = preprocess.crop_resize_records(frames, box, (224, 224), stem, n)
for _ in range(3): preprocess.crop_resize_records(frames, box, (224, 224), stem, n, out=rec)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): preprocess.crop_resize_records(frames, box, (224, 224), stem, n, out=rec)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100
print("abl=%s: %.0f us per %d clips = %.2f us per clip; %.2f TB/s of (2.36 MB in + 9.63 MB out) per clip" % (os.environ.get("TEDSPAD_CROPTP_ABL", "0"), us, n, us / n, n * 12.0e6 / us / 1e6))
