import torch, sys, contextlib, io
sys.path.insert(0, '.')
from ted_spad_amd.model_loaders import load_fa_model
from ted_spad_amd.synth import synth_state_dict, synth_tensor
with contextlib.redirect_stdout(io.StringIO()):
    fa = load_fa_model(arch='unet')
fa.load_state_dict(synth_state_dict(fa.state_dict(), 0)); fa = fa.cuda().eval()
x = synth_tensor(0, 'bigframes', (720, 3, 224, 224), device='cuda')      # 720 x 224^2 x 64 ch = 2.3e9 elements at level 0: chunked
with torch.no_grad():
    y = fa(x)
    y2 = torch.cat([fa(x[:360]), fa(x[360:])])
torch.cuda.synchronize()
print('chunked == halves:', bool(torch.equal(y, y2)), tuple(y.shape), float(y.mean()))
