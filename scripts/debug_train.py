import os, sys, torch, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from oracle import i3res50_ref, unet_ref
from ted_spad_amd.model_loaders import load_fa_model, load_ft_model
from ted_spad_amd.synth import synth_state_dict, synth_tensor
from ted_spad_amd.train_nets import I3DTrainer, UNetTrainer
def rel(a,b): a=a.double().cpu(); b=b.double().cpu(); return float((a-b).norm()/b.norm())
which = sys.argv[1] if len(sys.argv)>1 else 'both'
q16 = lambda t, kind: t.to(torch.float16).float()
if which in ('i3d','both'):
    ft = load_ft_model('largei3d', num_classes=102); sd = synth_state_dict(ft.state_dict(), 0); ft.load_state_dict(sd); ft = ft.cuda().eval(); ft.i3d.drop_p=0
    x = synth_tensor(0, 'dbgx', (2,3,16,32,32)).requires_grad_()
    i3 = {k[4:]: v for k, v in sd.items() if k.startswith('i3d.')}
    fq = i3res50_ref.extract_features(x, i3, q=q16).flatten(1)
    pred = torch.nn.functional.linear(fq, sd['i3d.fc.weight'], sd['i3d.fc.bias']); feat = i3res50_ref.mlp(fq, sd)
    dp = synth_tensor(0,'dp',tuple(pred.shape),-1,1); dfe = synth_tensor(0,'df',tuple(feat.shape),-1,1)
    (pred*dp).sum().backward(retain_graph=True); gx_pred = x.grad.clone(); x.grad=None
    (feat*dfe).sum().backward(); gx_feat = x.grad.clone()
    tr = I3DTrainer(ft)
    p, f, tape = tr.forward(x.detach().cuda(), 'eval')
    print('fwd pred', rel(p, pred.detach()), 'feat', rel(f, feat.detach()))
    dx1 = tr.backward(tape, dp.cuda(), None).clone()
    print('dx from pred', rel(dx1, gx_pred))
    dx2 = tr.backward(tape, None, dfe.cuda()).clone()
    print('dx from feat', rel(dx2, gx_feat))
if which in ('unet','both'):
    fa = load_fa_model(arch='unet'); sd = synth_state_dict(fa.state_dict(), 0)
    for k in sd:
        if k.rsplit('.',1)[0]+'.running_mean' in sd and k.endswith('.bias'): sd[k] = torch.full_like(sd[k], 4.0)
    fa.load_state_dict(sd); fa = fa.cuda().train()
    x = synth_tensor(0, 'dbgu', (6,3,32,32))
    sdg = {k:(v.clone().requires_grad_() if v.is_floating_point() and 'running' not in k else v) for k,v in sd.items()}
    y = unet_ref.forward(x, sdg, train=True)
    dy = synth_tensor(0,'dyu',tuple(y.shape),-1,1)
    (y*dy).sum().backward()
    tr = UNetTrainer(fa)
    yy, tape = tr.forward(x.cuda())
    print('unet fwd', rel(yy, y.detach()))
    tr.backward(tape, dy.cuda())
    tr.flush_grads()
    errs = {k: rel(p.grad, sdg[k].grad) for k,p in fa.named_parameters() if float(sdg[k].grad.norm())>1e-4}
    for k,v in errs.items(): print('%-45s %.3e  |g|=%.3e' % (k, v, float(sdg[k].grad.norm())))
