"""Timing of the I3Res50 stem + maxpool1 at bench size: pixel-pair halo stem + (2,3,3) max-pool against the persistent
time-channels-last stem (csrc/conv_stem_pt.hip) + (1,3,3) max-pool. Usage: python scripts/stem_pt_probe.py [clips]"""
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
pc = E.PackedConv(wgt, scale, shift, stride=(2, 2, 2), dtype="f16", device="cuda", pair_w=3)
st = E.StemPT(wgt, scale, shift, dtype="f16", device="cuda")


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


for _ in range(50):    # let the tuner settle on the old stem
    a = E.clip_to_act(x, cpad=4, dtype="f16")
    y_old = pc(a, pads=(2, 3, pc.pair_pw), pads_back=(2, 3, 1))
p_old = E.maxpool(y_old, (2, 3, 3), (2, 2, 2))
print("old: layout %.0f us, stem %.0f us, pool %.0f us" % (
    timed(lambda: E.clip_to_act(x, cpad=4, dtype="f16")), timed(lambda: pc(a, pads=(2, 3, pc.pair_pw), pads_back=(2, 3, 1))),
    timed(lambda: E.maxpool(y_old, (2, 3, 3), (2, 2, 2)))))
del y_old
xtc = st.layout(x)
for v in (0, 2):
    y = st.conv(xtc, variant=v)
    p_new = E.maxpool(y, (1, 3, 3), (1, 2, 2))
    d = (p_new.buf.float() - p_old.buf.float()).abs()
    print("variant %d: layout %.0f us, stem %.0f us, pool %.0f us; vs old pooled: max abs diff %.3g, mismatching %.4f %%" % (
        v, timed(lambda: st.layout(x)), timed(lambda: st.conv(xtc, variant=v)), timed(lambda: E.maxpool(y, (1, 3, 3), (1, 2, 2))),
        float(d.max()), 100.0 * float((d > 0).float().mean())))
    f = st.conv_pool(xtc, variant=v)
    print("variant %d, pool fused: stem + pool %.0f us; identical to the separate pool: %s" % (
        v, timed(lambda: st.conv_pool(xtc, variant=v)), bool(torch.equal(f.buf, p_new.buf))))
    del f
f16 = st.conv_pool(xtc, variant=6)
print("variant 6 (16x16x32 MFMAs), pool fused: %.0f us; max |diff| vs variant 2: %.3g" % (
    timed(lambda: st.conv_pool(xtc, variant=6)), float((f16.buf.float() - st.conv_pool(xtc, variant=2).buf.float()).abs().max())))
del f16
for nwg in (256, 512):
    st.nwg = nwg
    print("nwg %d: stem %.0f us (4 waves), %.0f us (8 waves)" % (nwg, timed(lambda: st.conv(xtc, variant=0)), timed(lambda: st.conv(xtc, variant=2))))
for dbg, what in ((1, "no halo DMA"), (2, "no stores"), (3, "no DMA, no stores"), (4, "no MFMA"), (7, "loop skeleton only")):
    print("ablation %s: %.0f us (4 waves), %.0f us (8 waves); pool fused %.0f us (8 waves)" % (
        what, timed(lambda: st.conv(xtc, variant=0 | (dbg << 8))), timed(lambda: st.conv(xtc, variant=2 | (dbg << 8))),
        timed(lambda: st.conv_pool(xtc, variant=2 | (dbg << 8)))))
