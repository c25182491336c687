"""Time the I3Res50 / InceptionI3d stems under the tile configurations that apply (9: halo-direct, 20: two-frame halo-direct)."""
import contextlib, io, os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ted_spad_amd import engine as E, _lib
from ted_spad_amd.model_loaders import load_ft_model
from ted_spad_amd.synth import synth_state_dict, synth_clips
with contextlib.redirect_stdout(io.StringIO()):
    ft = load_ft_model('largei3d', num_classes=102)
ft.load_state_dict(synth_state_dict(ft.state_dict(), 0)); ft = ft.cuda().eval()
for batch in (75, 225):
    a = E.clip_to_act(synth_clips(0, batch, (3, 16, 224, 224), device='cuda'), cpad=4)
    st = ft.i3d.packed()["stem"]
    call = lambda: st(a, pads=(2, 3, st.pair_pw), pads_back=(2, 3, 1))
    outs = {}
    for cfg in (9, 20, 21, 29, 30):
        E.FORCE_TILE_CFG = cfg
        try:
            o = call()
        except _lib.TedSpadHipError as e:
            print('cfg', cfg, 'n/a', e); continue
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): call()
        e1.record(); torch.cuda.synchronize()
        outs[cfg] = o.buf.clone()
        print('batch', batch, 'cfg', cfg, '%.1f us' % (e0.elapsed_time(e1) / 10 * 1e3))
    if 20 in outs: print('20 == 9:', bool(torch.equal(outs[9], outs[20])))
    if 29 in outs: print('29 == 9:', bool(torch.equal(outs[9], outs[29])), ' 30 == 21:', bool(torch.equal(outs[21], outs[30])) if 30 in outs else None)
    if 21 in outs: print('21 vs 9: max |diff| %.3e, differing %.4f %%' % (float((outs[21].float() - outs[9].float()).abs().max()), 100 * float((outs[21] != outs[9]).float().mean())))
    E.FORCE_TILE_CFG = None
