"""Time of the fp32 head linears on the shapes of the cfg3 step (fb's mlp: forward, data gradient, weight gradient = short reduction over the batch), with a checksum
(two builds of the library must print the same bits). Usage: python scripts/linear_probe.py"""
import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ted_spad_amd import head
for B, K, N in ((24, 2048, 2048), (24, 2048, 128), (2048, 24, 2048), (128, 24, 2048), (2048, 24, 128), (102, 24, 2048), (2048, 12, 2048)):
    g = torch.Generator().manual_seed(B * 131 + K + N)
    x, w, b = torch.randn(B, K, generator=g).cuda(), (torch.randn(N, K, generator=g) / K ** 0.5).cuda(), torch.randn(N, generator=g).cuda()
    y = head.linear(x, w, b); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): y = head.linear(x, w, b)
    e1.record(); torch.cuda.synchronize()
    print('B %5d K %5d N %5d: %7.1f us  checksum %.9e' % (B, K, N, e0.elapsed_time(e1) * 50, float(y.double().sum())))
