"""Which tile configuration the in-context tuner settles on for every conv geometry of a 375-clip I3Res50 forward. Usage: python scripts/tile_picks.py [clips]"""
import os, sys, io, contextlib, collections, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ted_spad_amd import engine as E
from ted_spad_amd.model_loaders import load_ft_model
from ted_spad_amd.synth import synth_clips, synth_state_dict
n = int(sys.argv[1]) if len(sys.argv) > 1 else 375
with contextlib.redirect_stdout(io.StringIO()):
    ft = load_ft_model("largei3d", num_classes=102)
ft.load_state_dict(synth_state_dict(ft.state_dict(), 0)); ft = ft.cuda().eval()
x = torch.cat([synth_clips(0, min(25, n - i), (3, 16, 224, 224), device="cuda", first=i) for i in range(0, n, 25)])
with torch.no_grad():
    for i in range(120):
        ft.i3d.extract_features(x)
        if i > 48 and not E.tuning_pending():
            break
tab = E.export_tile_table(ft.i3d.packed())
cnt = collections.Counter()
for name, d in sorted(tab.items()):
    for key, cfg in d.items():
        cnt[cfg] += 1
        print("%-28s cfg %2d  %s" % (name, cfg, key[:8]))
print("picks per tile_cfg:", dict(sorted(cnt.items())))
