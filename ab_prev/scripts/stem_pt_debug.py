import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle.conv_ref import conv_cl
from ted_spad_amd import engine as E
from ted_spad_amd.synth import synth_tensor
shape = (2, 3, 16, 32, 32)
n, c, t, h, w = shape
clip = synth_tensor(5, "ptclip%d" % h, shape).half().float()
wgt = synth_tensor(5, "tuw", (64, c, 5, 7, 7), -0.1, 0.1).half().float()
scale, shift = synth_tensor(5, "tus", (64,), 0.5, 1.5), synth_tensor(5, "tub", (64,), -0.3, 0.3)
full = conv_cl(clip.permute(0, 2, 3, 4, 1), wgt, scale, shift, (2, 2, 2), (2, 3, 3), (2, 3, 3))
tp = full.shape[1] // 2
ref = torch.maximum(full[:, 0:2 * tp:2], full[:, 1:2 * tp:2])
st = E.StemPT(wgt, scale, shift, dtype="f16", device="cuda")
xtc = st.layout(clip.cuda())
big = synth_tensor(5, "ptclip%d" % h, shape)
lay = xtc.float().cpu()
exp = torch.zeros(n, h, w, 64); exp[..., 6:54] = clip.permute(0, 3, 4, 2, 1).reshape(n, h, w, 48)
print("layout equal:", torch.equal(lay, exp))
for v in (0, 1):
    got = st.conv(xtc, t, variant=v).buf.float().cpu()
    bad = ~((got - ref).abs() <= 2.0 ** -10 * ref.abs() + 2e-3)
    print("variant", v, "bad", int(bad.sum()), "of", bad.numel(), "nan", int(torch.isnan(got).sum()))
    print(" by n,tp:", bad.sum(dim=(2, 3, 4)).tolist())
    print(" by row:", bad.sum(dim=(0, 1, 3, 4)).tolist())
    print(" by col:", bad.sum(dim=(0, 1, 2, 4)).tolist())
    print(" by ch:", bad.sum(dim=(0, 1, 2, 3)).tolist())
    idx = bad.nonzero()[:8]
    for i in idx:
        i = tuple(i.tolist())
        print("  ", i, float(got[i]), float(ref[i]), float(full[i[0], 2 * i[1], i[2], i[3], i[4]]), float(full[i[0], 2 * i[1] + 1, i[2], i[3], i[4]]))
