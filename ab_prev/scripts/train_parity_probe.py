"""Phase-2 gradient parity at the cfg3 shape against the fp32 oracle for several static loss scales: does the error come from 16-bit
activation GRADIENTS (subnormal / flushed values) rather than from ReLU flips of the forward? Usage: python scripts/train_parity_probe.py"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import train_step_ref
from ted_spad_amd.synth import synth_train_video
from ted_spad_amd.train_step import AnonymizerTrainStep
from test_hip_train_step import _models

shape = (8, 48, 3, 112, 112) if len(sys.argv) < 2 else eval(sys.argv[1])
video = synth_train_video(0, "train_cfg3", shape)
labels = torch.tensor([5, 77, 101, 1, 33, 60, 12, 90][:shape[0]])
fa, ft, sd_u, sd_l = _models()
torch.set_num_threads(32)
ref_l, ref_g = train_step_ref.phase2(video, labels, sd_u, sd_l)
print("oracle loss_ft %.6f" % ref_l["loss_ft"])
for ls in (1.0, 64.0, 4096.0, 262144.0):
    fa, ft, _, _ = _models()
    step = AnonymizerTrainStep(fa, ft, loss_scale=ls)
    step.opt_ft = torch.optim.SGD(ft.parameters(), lr=0.0)
    out = step.step_ft(video.cuda(), labels.cuda())
    errs, cos = [], []
    for k, p in ft.named_parameters():
        g, r = p.grad.detach().cpu().double().flatten(), ref_g[k].double().flatten()
        if float(r.norm()) > 1e-4:
            errs.append(float((g - r).norm() / r.norm())); cos.append(float(g @ r / (g.norm() * r.norm())))
    print("loss_scale %-9g loss_ft %.6f skipped %s | grads: median rel-L2 %.4f, worst %.4f, median cos %.4f, min cos %.4f" % (
        ls, out["loss_ft"], out["skipped"], np.median(errs), max(errs), np.median(cos), min(cos)))
