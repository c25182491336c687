"""Timing of the pool-fused 16x16x32 stem at bench size: layout pass + kernel on the records against the kernel reading the fp32 clip itself
(tedspad_stem_pt_pool_clip_fwd), with the timing ablations of both. Usage: python scripts/stem_clip_probe.py [clips]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ted_spad_amd import engine as E
from ted_spad_amd.synth import synth_clips, synth_tensor

n = int(sys.argv[1]) if len(sys.argv) > 1 else 225
x = torch.cat([synth_clips(0, min(25, n - i), (3, 16, 224, 224), device="cuda", first=i) for i in range(0, n, 25)])
wgt = synth_tensor(5, "w", (64, 3, 5, 7, 7), -0.05, 0.05).cuda()
scale, shift = synth_tensor(5, "s", (64,), 0.5, 1.5).cuda(), synth_tensor(5, "b", (64,), -0.3, 0.3).cuda()
st = E.StemPT(wgt, scale, shift, dtype="f16", device="cuda")


def timed(fn, reps=7):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


xtc = st.layout(x)
ref = st.conv_pool(xtc, variant=6)
got = st.conv_pool_clip(x)
print("identical:", bool(torch.equal(ref.buf, got.buf)))
print("layout %.0f us, stem on records %.0f us, stem on the fp32 clip %.0f us" % (timed(lambda: st.layout(x)), timed(lambda: st.conv_pool(xtc, variant=6)),
                                                                                  timed(lambda: st.conv_pool_clip(x))))
for dbg, what in ((1, "no halo loads"), (2, "no stores"), (4, "no MFMA"), (5, "no MFMA, no loads")):
    print("ablation %-20s: records %.0f us, fp32 clip %.0f us" % (what, timed(lambda: st.conv_pool(xtc, variant=6 | (dbg << 8))),
                                                                    timed(lambda: st.conv_pool_clip(x, variant=dbg << 8))))
