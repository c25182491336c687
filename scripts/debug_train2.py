import os, sys, torch, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from oracle import i3res50_ref
from ted_spad_amd.model_loaders import load_ft_model
from ted_spad_amd.synth import synth_state_dict, synth_tensor
from ted_spad_amd.train_nets import I3DTrainer
def rel(a,b): a=a.double().cpu().flatten(); b=b.double().cpu().flatten(); return float((a-b).norm()/b.norm()), float(a@b/(a.norm()*b.norm()))
ft = load_ft_model('largei3d', num_classes=102); sd = synth_state_dict(ft.state_dict(), 0); ft.load_state_dict(sd); ft = ft.cuda().train(); ft.i3d.drop_p=0
x = synth_tensor(0, 'dbgx', (4,3,16,64,64)) * (torch.arange(1,5).float()/4).view(4,1,1,1,1)
sdg = {k:(v.clone().requires_grad_() if v.is_floating_point() and 'running' not in k else v) for k,v in sd.items()}
pred, feat = i3res50_ref.wrapper_forward(x, sdg, train=True)
dp = synth_tensor(0,'dp',tuple(pred.shape),-1,1); dfe = synth_tensor(0,'df',tuple(feat.shape),-1,1)
((pred*dp).sum() + (feat*dfe).sum()).backward()
tr = I3DTrainer(ft)
p, f, tape = tr.forward(x.cuda(), 'train')
print('fwd pred', rel(p, pred.detach()), 'feat', rel(f, feat.detach()))
tr.backward(tape, dp.cuda(), dfe.cuda())
tr.flush_grads()
keys = [k for k,_ in ft.named_parameters()]
for k in reversed(keys):
    if float(sdg[k].grad.norm()) > 1e-5 and ('layer4' in k or 'layer3.5' in k or 'mlp' in k or 'fc' in k or k.startswith('i3d.conv1') or k.startswith('i3d.bn1') or 'layer1.0' in k):
        print('%-40s rel %.3e cos %.4f' % ((k,) + rel(dict(ft.named_parameters())[k].grad, sdg[k].grad)))
