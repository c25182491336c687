"""ORACLE (test infrastructure, never shipped as product): CPU restatement of the
feature-extraction drivers and of the downstream `.npy` consumer.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Pinned by tests/test_oracle_golden.py (Q1/Q2 index maps, `.npy` header/layout and the
consumer output shape captured from the reference's own code paths).

Follows (reference file:line):
  * extract_features            feature_extraction/st_feature_extraction.py:16-37
  * inline extraction loop      feature_extraction/dali_extraction.py:151-182
  * quirk Q1 (reshape, not permute)  dali_extraction.py:171-173, st_feature_extraction.py:24-26
  * quirk Q2 (training pseudo-images) anonymization_training/train_anonymizer.py:87-92
  * process_feat                anomaly_detection_mgfn/utils/utils.py:34-42
  * Dataset.__getitem__         anomaly_detection_mgfn/datasets/dataset.py:51-100
"""
import numpy as np
import torch


def q1_feed(clip_t_c_hw, fa=None):
    """clip: (1, 16, 3, H, W) as the loaders deliver it (frames, then colour).
    `view(-1,3,H,W)` hands TRUE RGB frames to fa; the result is `.reshape(1,3,16,H,W)`,
    a reinterpretation of (T,C) memory as (C,T): ft's input [0,ch,t] is frame
    (ch*16+t)//3, colour (ch*16+t)%3."""
    b, t, c, h, w = clip_t_c_hw.shape
    frames = clip_t_c_hw.reshape(-1, c, h, w)
    if fa is not None:
        frames = fa(frames)
    return frames.reshape(b, c, t, h, w)


def q1_index_map(t=16, c=3):
    """(frame, colour) that lands at ft-input position [ch, tt]."""
    ch, tt = np.meshgrid(np.arange(c), np.arange(t), indexing="ij")
    flat = ch * t + tt
    return flat // c, flat % c


def q2_feed(video_b_c_t_hw, fa=None):
    """train_anonymizer.py:87-92: (B,3,48,H,W).reshape(-1,3,H,W) -> fa -> reshape back."""
    b, c, t, h, w = video_b_c_t_hw.shape
    imgs = video_b_c_t_hw.reshape(-1, c, h, w)
    if fa is not None:
        imgs = fa(imgs)
    return imgs.reshape(b, c, t, h, w)


def extract_video(clips, ft_extract, fa=None, layout="reference"):
    """clips: list/tensor of (16,3,H,W) fp32 [0,1] clips of ONE video.
    Returns float64 (T, F) exactly as `np.save` receives it (np.zeros default dtype,
    st_feature_extraction.py:94)."""
    rows = []
    for clip in clips:
        x = clip.unsqueeze(0)  # (1,16,3,H,W)
        if layout == "reference":
            x = q1_feed(x, fa)
        else:  # 'permute': the geometrically meaningful feed
            if fa is not None:
                x = fa(x.reshape(-1, *x.shape[2:])).reshape(x.shape)
            x = x.permute(0, 2, 1, 3, 4)
        with torch.no_grad():
            f = ft_extract(x)
        rows.append(f.squeeze().cpu().numpy())
    out = np.zeros((len(rows), rows[0].shape[-1]))
    for i, r in enumerate(rows):
        out[i] = r
    return out


def process_feat(feat, length):
    """utils/utils.py:34-42: 32-segment mean pooling with the linspace(int) boundaries."""
    new_feat = np.zeros((length, feat.shape[1]), np.float32)
    r = np.linspace(0, len(feat), length + 1, dtype=int)
    for i in range(length):
        if r[i] != r[i + 1]:
            new_feat[i] = feat[r[i]:r[i + 1]].mean(0)
        else:
            new_feat[i] = feat[r[i]]
    return new_feat


def mgfn_getitem(npy_path, test_mode=False, seg_length=32):
    """dataset.py:51-100 with its argparse/global state removed."""
    feats = np.array(np.load(npy_path, allow_pickle=True), dtype=np.float32)
    if feats.ndim < 3:
        feats = np.expand_dims(feats, 1)  # (T, 1, F)
    if test_mode:
        mag = np.linalg.norm(feats, axis=2)[:, :, None]
        return np.concatenate((feats, mag), 2)  # (T, ncrops, F+1)
    feats = feats.transpose(1, 0, 2)  # (ncrops, T, F)
    div = np.stack([process_feat(f, seg_length) for f in feats]).astype(np.float32)
    mag = np.linalg.norm(div, axis=2)[:, :, None].astype(np.float32)
    return np.concatenate((div, mag), 2)  # (ncrops, 32, F+1)
