"""-m gpu: fused loss kernels (value + gradients) against the golden vectors captured from
the reference's own NTXentLoss / TripletMarginLoss / CrossEntropyLoss and the float64 oracle."""
import numpy as np
import pytest
import torch

from conftest import rel_l2
from ted_spad_amd.synth import synth_tensor

pytestmark = pytest.mark.gpu


def _unit(name, shape):
    return torch.nn.functional.normalize(synth_tensor(0, name, shape, -1, 1), dim=1)


def test_ntxent_golden(golden):
    from ted_spad_amd.losses import NTXentLoss
    zi = _unit("ntx_zi", (12, 128)).cuda().requires_grad_()
    zj = _unit("ntx_zj", (12, 128)).cuda().requires_grad_()
    l = NTXentLoss("cuda", 12, 0.1, False)(zi, zj)
    l.backward()
    assert abs(l.item() - golden["ntxent_value"][0]) < 2e-5 * abs(golden["ntxent_value"][0])
    assert rel_l2(zi.grad.cpu(), golden["ntxent_grad_zi"]) < 1e-4
    assert rel_l2(zj.grad.cpu(), golden["ntxent_grad_zj"]) < 1e-4


@pytest.mark.parametrize("n,c,cos", [(1, 2, False), (4, 128, False), (12, 128, True), (32, 256, False), (17, 64, True)])
def test_ntxent_vs_oracle(n, c, cos):
    from oracle import losses_ref
    from ted_spad_amd.losses import NTXentLoss
    zi0 = synth_tensor(1, "zi%d" % n, (n, c), -1, 1) * 0.3
    zj0 = synth_tensor(1, "zj%d" % n, (n, c), -1, 1) * 0.3
    zi, zj = zi0.cuda().requires_grad_(), zj0.cuda().requires_grad_()
    l = NTXentLoss("cuda", n, 0.1, cos)(zi, zj)
    (l * 2.0).backward()  # also checks scaling by the incoming gradient
    ref = losses_ref.nt_xent_np(zi0.numpy(), zj0.numpy(), 0.1, use_cosine=cos)
    assert abs(l.item() - ref) < 1e-4 * max(1.0, abs(ref))
    a, b = zi0.double().requires_grad_(), zj0.double().requires_grad_()
    if cos:
        lr = losses_ref.nt_xent_torch(torch.nn.functional.normalize(a, dim=1, eps=1e-8), torch.nn.functional.normalize(b, dim=1, eps=1e-8), 0.1)
    else:
        lr = losses_ref.nt_xent_torch(a, b, 0.1)
    (lr * 2.0).backward()
    if n > 1 or not cos:
        assert rel_l2(zi.grad.cpu(), a.grad) < 2e-4
        assert rel_l2(zj.grad.cpu(), b.grad) < 2e-4


def test_triplet_and_ce_golden(golden):
    from ted_spad_amd.losses import CrossEntropyLoss, TripletMarginLoss
    a, p, n = (_unit("trip_" + s, (8, 128)).cuda().requires_grad_() for s in "apn")
    l = TripletMarginLoss(margin=1)(a, p, n)
    l.backward()
    assert abs(l.item() - golden["triplet_value"][0]) < 1e-5
    for t, k in ((a, "a"), (p, "p"), (n, "n")):
        assert rel_l2(t.grad.cpu(), golden["triplet_grad_" + k]) < 1e-5
    lg = synth_tensor(0, "ce_logits", (8, 102), -3, 3).cuda().requires_grad_()
    lab = torch.from_numpy(golden["ce_labels"]).cuda()
    lc = CrossEntropyLoss()(lg, lab)
    lc.backward()
    assert abs(lc.item() - golden["ce_value"][0]) < 1e-5
    assert rel_l2(lg.grad.cpu(), golden["ce_grad"]) < 1e-5


def test_triplet_inactive_rows_and_margin():
    from oracle import losses_ref
    from ted_spad_amd.losses import TripletMarginLoss
    a = synth_tensor(2, "ta", (6, 128), -1, 1)
    p = a + synth_tensor(2, "tp", (6, 128), -1, 1) * 0.05
    n = a + synth_tensor(2, "tn", (6, 128), -1, 1) * 5.0   # far negatives: hinge inactive
    n[3:] = a[3:] + synth_tensor(2, "tn2", (3, 128), -1, 1) * 0.05   # near negatives: active
    A, P, Nn = (t.cuda().requires_grad_() for t in (a, p, n))
    l = TripletMarginLoss(margin=1)(A, P, Nn)
    l.backward()
    ad, pd, nd = (t.double().requires_grad_() for t in (a, p, n))
    lr = losses_ref.triplet_torch(ad, pd, nd)
    lr.backward()
    assert abs(l.item() - lr.item()) < 1e-5
    assert float(A.grad[:3].abs().max()) == 0.0
    assert rel_l2(A.grad.cpu(), ad.grad) < 1e-4 and rel_l2(Nn.grad.cpu(), nd.grad) < 1e-4 and rel_l2(P.grad.cpu(), pd.grad) < 1e-4
