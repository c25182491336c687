import torch
def t(f, n=20):
    f(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e-3
for mb in (256, 1024, 4096):
    x=torch.empty(mb*1024*1024//2, dtype=torch.float16, device='cuda').normal_()
    y=torch.empty_like(x)
    b=x.numel()*2
    print(mb,'MB copy  %.2f TB/s (r+w)' % (2*b/t(lambda: y.copy_(x))/1e12), ' fill %.2f TB/s' % (b/t(lambda: y.fill_(1))/1e12), ' read(sum) %.2f TB/s' % (b/t(lambda: x.float().sum() if False else torch.sum(x, dtype=torch.float32))/1e12))
