"""SURVEY.md §7's bf16 plan, emulated on the CPU oracle (oracle/i3res50_ref.py's rounding hook q(tensor, kind)): what feature rel-L2 would I3Res50 reach with bf16 MFMA
operands (weights and conv inputs rounded to bf16) while the RESIDUAL STREAM (the tensor every bottleneck adds to) is kept in f16 or fp32? Against the exact fp32 path, on
synthetic weights / clips; the all-f16 and all-bf16 rows are what the two built dtypes measure on the device (4.1e-4 / 3.0e-3).   python scripts/bf16_residual_experiment.py [hw]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import i3res50_ref as R
from ted_spad_amd.model_loaders import load_ft_model
from ted_spad_amd.synth import synth_clips, synth_state_dict
import contextlib, io
hw = int(sys.argv[1]) if len(sys.argv) > 1 else 112
torch.set_num_threads(8)
with contextlib.redirect_stdout(io.StringIO()):
    ft = load_ft_model("largei3d", num_classes=102)
sd = {k[4:]: v for k, v in synth_state_dict(ft.state_dict(), 0).items() if k.startswith("i3d.")}
x = synth_clips(0, 2, (3, 16, hw, hw))
def mk(op, res):
    def q(t, kind):
        d = res if kind == "res" else op
        return t if d is None else t.to(d).float()
    return q
with torch.no_grad():
    ref = R.extract_features(x, sd).flatten(1)
    for name, op, res in (("f16 operands, f16 residual stream (the device's f16 mode)", torch.float16, torch.float16),
                          ("bf16 operands, bf16 residual stream (the device's bf16 mode)", torch.bfloat16, torch.bfloat16),
                          ("bf16 operands, f16 residual stream", torch.bfloat16, torch.float16),
                          ("bf16 operands, fp32 residual stream", torch.bfloat16, None),
                          ("f16 operands, fp32 residual stream", torch.float16, None)):
        f = R.extract_features(x, sd, q=mk(op, res)).flatten(1)
        rel = ((f - ref).norm(dim=1) / ref.norm(dim=1)).max().item()
        print("%-62s max rel-L2 over 2 clips @%d^2: %.2e" % (name, hw, rel), flush=True)
