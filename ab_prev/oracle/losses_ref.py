"""ORACLE (test infrastructure, never shipped as product): CPU restatement of the three
losses of the anonymizer training step, in float64 numpy (values) and torch float64
autograd (gradients).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Parity is PINNED by tests/test_oracle_golden.py against values + gradients captured by
running the reference's own `NTXentLoss`, `nn.TripletMarginLoss` and
`nn.CrossEntropyLoss` (tests/golden/make_golden.py).

Follows (reference file:line):
  * NTXentLoss.forward              aux_code/nt_xent_original.py:49-70
    mask of "same representation"   aux_code/nt_xent_original.py:26-32
    dot similarity                  aux_code/nt_xent_original.py:35-40
  * nn.TripletMarginLoss(margin=1)  anonymization_training/train_anonymizer.py:349-350,115
    (torch semantics: d(x,y) = ||x - y + 1e-6||_2, mean over the batch)
  * nn.CrossEntropyLoss()           anonymization_training/train_anonymizer.py:347,107
  * loss algebra of both phases     anonymization_training/train_anonymizer.py:116-119,182-183
"""
import numpy as np
import torch


def nt_xent_np(zis, zjs, temperature=0.1, use_cosine=False):
    """Literal restatement: R = cat[zjs, zis]; S = R R^T; positives = diagonals +-N;
    negatives = everything but the main and +-N diagonals; logits = [pos | neg] / T;
    CE(sum, label 0) / 2N."""
    zis = np.asarray(zis, np.float64)
    zjs = np.asarray(zjs, np.float64)
    n = zis.shape[0]
    r = np.concatenate([zjs, zis], 0)
    if use_cosine:
        nr = np.maximum(np.linalg.norm(r, axis=1, keepdims=True), 1e-8)
        s = (r / nr) @ (r / nr).T
    else:
        s = r @ r.T
    pos = np.concatenate([np.diag(s, n), np.diag(s, -n)]).reshape(2 * n, 1)
    keep = np.ones((2 * n, 2 * n), bool)
    idx = np.arange(2 * n)
    keep[idx, idx] = False
    keep[idx, (idx + n) % (2 * n)] = False
    neg = s[keep].reshape(2 * n, 2 * n - 2)
    logits = np.concatenate([pos, neg], 1) / temperature
    m = logits.max(1, keepdims=True)
    lse = m[:, 0] + np.log(np.exp(logits - m).sum(1))
    return float((lse - logits[:, 0]).sum() / (2 * n))


def nt_xent_torch(zis, zjs, temperature=0.1):
    """Closed form: CE(S/T with the main diagonal removed,
    target (i+N) mod 2N), mean over the 2N rows; differentiable, float64."""
    n = zis.shape[0]
    r = torch.cat([zjs, zis], 0)
    s = (r @ r.t()) / temperature
    s = s.masked_fill(torch.eye(2 * n, dtype=torch.bool), float("-inf"))
    tgt = (torch.arange(2 * n) + n) % (2 * n)
    return (torch.logsumexp(s, 1) - s[torch.arange(2 * n), tgt]).mean()


def triplet_np(a, p, n, margin=1.0, eps=1e-6):
    a, p, n = (np.asarray(t, np.float64) for t in (a, p, n))
    dp = np.sqrt(((a - p + eps) ** 2).sum(1))
    dn = np.sqrt(((a - n + eps) ** 2).sum(1))
    return float(np.maximum(dp - dn + margin, 0.0).mean())


def triplet_torch(a, p, n, margin=1.0, eps=1e-6):
    dp = ((a - p + eps) ** 2).sum(1).sqrt()
    dn = ((a - n + eps) ** 2).sum(1).sqrt()
    return torch.clamp(dp - dn + margin, min=0).mean()


def cross_entropy_np(logits, labels):
    logits = np.asarray(logits, np.float64)
    m = logits.max(1, keepdims=True)
    lse = m[:, 0] + np.log(np.exp(logits - m).sum(1))
    return float((lse - logits[np.arange(len(labels)), np.asarray(labels)]).mean())


def cross_entropy_torch(logits, labels):
    return (torch.logsumexp(logits, 1) - logits[torch.arange(logits.shape[0]), labels]).mean()


def loss_ft(logits, labels, f1, f2, f3, temporal_loss_weight=0.1, margin=1.0):
    """train_anonymizer.py:107,115-116 / :176,182-183: CE + w_t * Triplet."""
    return cross_entropy_torch(logits, labels) + temporal_loss_weight * triplet_torch(f1, f2, f3, margin)


def loss_fa(l_fb, l_ft, fb_loss_weight=1.0, ft_loss_weight=0.7):
    """train_anonymizer.py:119: fa MAXIMISES the privacy NT-Xent, minimises the utility loss."""
    return -fb_loss_weight * l_fb + ft_loss_weight * l_ft
