"""Feature-extraction drivers: the build's counterpart of
`feature_extraction/st_feature_extraction.py:16-37` (extract_features) and of the inline loop
`feature_extraction/dali_extraction.py:151-182`, batched for MI355X.

Same contract: per video a float64 C-order `(T, F)` `.npy` (np.zeros default dtype,
st_feature_extraction.py:94), one row per 16-frame clip, file name = video
basename without extension; the optional anonymizer feed reproduces the reference's
reshape-not-permute quirk (Q1) by default. `(T, ncrops, F)` is the other layout the MGFN
loader accepts (anomaly_detection_mgfn/datasets/dataset.py:70-89) and is what the 10-crop
config writes.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import sharding


def _extract_fn(ft_model):
    # dali_extraction.py:175-178: ft_model.extract_features, else ft_model.i3d.extract_features
    return ft_model.extract_features if hasattr(ft_model, "extract_features") else ft_model.i3d.extract_features


def feature_width(ft_model) -> int:
    """F of the (n, F) rows `extract_features` returns: 2048 for I3Res50 / wrapper_i3d (large_i3d.py:262), 1024 for InceptionI3d
    (i3d.py:336-340). Taken from the model, not from a probe forward: an empty shard must not launch anything."""
    for m in (ft_model, getattr(ft_model, "i3d", None)):
        f = getattr(m, "feature_dim", None) if m is not None else None
        if f is not None:
            return int(f)
    raise AttributeError("extraction: the feature extractor must expose `feature_dim` (I3Res50 2048, InceptionI3d 1024)")


def feed(clips: torch.Tensor, fa_model=None, layout: str = "reference", fa_batch: int = 0) -> torch.Tensor:
    """clips: (B, 16, 3, H, W) fp32 in [0,1] as the loaders deliver them -> ft input (B, 3, 16, H, W).

    layout='reference' reproduces st_feature_extraction.py:24-26 / dali_extraction.py:171-173: the
    frames go through fa as true RGB images (view(-1,3,H,W)), and the result is RESHAPED to
    (B,3,16,H,W) -- a reinterpretation of (T,C) memory as (C,T). Without fa the
    reference would hand a 16-channel tensor to a 3-channel conv; the counterpart permutes.
    layout='permute' is the geometrically meaningful feed.
    fa_batch > 0: the anonymizer runs on `fa_batch` clips (16 x fa_batch frames) at a time -- its full-resolution activations are 40 x a clip -- while the
    encoder behind it gets all B clips in one forward (it needs 75+ clips to fill the chip); the frames are per-sample, so the result is the same.
    """
    b, t, c, h, w = clips.shape
    if fa_model is not None:
        if fa_batch and b > fa_batch:
            frames = torch.empty((b * t, c, h, w), dtype=torch.float32, device=clips.device)
            for i in range(0, b, fa_batch):
                k = min(fa_batch, b - i)
                frames[i * t:(i + k) * t] = fa_model(clips[i:i + k].reshape(-1, c, h, w))
        else:
            frames = fa_model(clips.reshape(-1, c, h, w))
        if layout == "reference":
            return frames.reshape(b, c, t, h, w)
        return frames.reshape(b, t, c, h, w).permute(0, 2, 1, 3, 4)
    return clips.permute(0, 2, 1, 3, 4)


_STREAMS = {}


@torch.no_grad()
def extract_clip_features(ft_model, clips_cthw: torch.Tensor, batch: int = 75, out: torch.Tensor = None, streams: int = 3) -> torch.Tensor:
    """clips_cthw: (n, 3, T, H, W) fp32 on the GPU -> (n, F) fp32 on the GPU, `batch` clips per forward.
    Batches alternate over `streams` HIP streams: the late, small-grid layers of one forward leave CUs idle that the
    next forward's early layers fill (+3-8 % clips/s measured; results are unchanged; 75 clips per forward quantise best on 256 CUs)."""
    fx = _extract_fn(ft_model)
    n = clips_cthw.shape[0]
    if out is None:
        out = torch.empty((n, feature_width(ft_model)), dtype=torch.float32, device=clips_cthw.device)
    if n == 0:                # an empty shard (T < world, sharding.shard_range): nothing to launch, the collective still runs
        return out
    dev = clips_cthw.device
    if dev.type != "cuda":    # host-logic tests drive the sharding with a stub extractor on CPU tensors: no streams there
        for i in range(0, n, batch):
            f = fx(clips_cthw[i:i + batch]).flatten(1)
            out[i:i + f.shape[0]] = f
        return out
    pool = _STREAMS.setdefault((dev, streams), [torch.cuda.Stream(device=dev) for _ in range(max(1, streams))])
    main = torch.cuda.current_stream(dev)
    for st in pool:
        st.wait_stream(main)
    from . import engine as E
    # forwards of at most `batch` clips, spread evenly over the streams (a 290-clip shard of a video split over 8 ranks: 145 + 145, not 290 on one stream)
    for j, (i, k) in enumerate(sharding.batch_plan(n, batch, len(pool))):
        # while the conv tile tuner is still timing candidates, stay on one stream (engine.tuning_pending)
        with torch.cuda.stream(pool[0 if E.tuning_pending() else j % len(pool)]):
            out[i:i + k] = fx(clips_cthw[i:i + k]).flatten(1)
    for st in pool:
        main.wait_stream(st)
    return out


@torch.no_grad()
def extract_anonymized_clip_features(ft_model, fa_model, clips: torch.Tensor, batch: int = 75, fa_batch: int = 75, layout: str = "reference",
                                    out: torch.Tensor = None, streams: int = 2) -> torch.Tensor:
    """clips: (n, 16, 3, H, W) fp32 on the GPU as the loaders deliver them -> fa -> `feed` layout -> ft.extract_features -> (n, F) fp32 on the GPU:
    `batch` clips per encoder forward, `fa_batch` per anonymizer forward, the batches alternating over `streams` HIP streams (as extract_clip_features does:
    one batch's prologues and epilogues under the other's MFMAs; results unchanged)."""
    fx = _extract_fn(ft_model)
    n = clips.shape[0]
    if out is None:
        out = torch.empty((n, feature_width(ft_model)), dtype=torch.float32, device=clips.device)
    if n == 0:
        return out
    dev = clips.device
    if dev.type != "cuda":
        for i in range(0, n, batch):
            f = fx(feed(clips[i:i + batch], fa_model, layout, fa_batch=fa_batch)).flatten(1)
            out[i:i + f.shape[0]] = f
        return out
    pool = _STREAMS.setdefault((dev, streams), [torch.cuda.Stream(device=dev) for _ in range(max(1, streams))])
    main = torch.cuda.current_stream(dev)
    for st in pool:
        st.wait_stream(main)
    from . import engine as E
    for j, i in enumerate(range(0, n, batch)):
        with torch.cuda.stream(pool[0 if E.tuning_pending() else j % len(pool)]):
            f = fx(feed(clips[i:i + batch], fa_model, layout, fa_batch=fa_batch)).flatten(1)
            out[i:i + f.shape[0]] = f
    for st in pool:
        main.wait_stream(st)
    return out


@torch.no_grad()
def extract_features(full_vid, vid_features, save_path, fa_model, ft_model, anonymized, segment=False,
                     batch: int = 75, layout: str = "reference", device="cuda", fa_batch: int = 75):
    """Drop-in for st_feature_extraction.py:16-37. full_vid: sequence of (16,3,H,W) clips;
    vid_features: preallocated float64 (len(full_vid), F) array that receives the rows;
    the array is saved to `save_path` with np.save (float64, C order)."""
    if segment:
        raise NotImplementedError("segment_features is dead + buggy code in the reference (SURVEY.md Q12)")
    for i in range(0, len(full_vid), batch):
        clips = torch.stack([c for c in full_vid[i:i + batch]]).to(device, non_blocking=True)
        x = feed(clips, fa_model if anonymized else None, layout, fa_batch=fa_batch)
        f = _extract_fn(ft_model)(x).flatten(1)
        vid_features[i:i + f.shape[0]] = f.cpu().numpy()
    warn_if_saturated(ft_model)
    np.save(save_path, vid_features)
    return vid_features


def warn_if_saturated(ft_model) -> int:
    """With `I3Res50.check_saturation` on (TEDSPAD_CHECK_SATURATION=1): reads the encoder's saturation counter and warns when an f16 activation store was clamped at
    +-65504 (the reference's fp32 / autocast run would carry an inf there): features of that video are not trustworthy. Returns the count (0 when the check is off)."""
    import warnings
    net = ft_model.i3d if hasattr(ft_model, "i3d") else ft_model
    if not getattr(net, "check_saturation", False) or not hasattr(net, "saturation_counts"):
        return 0
    sat, bad = net.saturation_counts(reset=True)
    if sat or bad:
        warnings.warn("I3Res50: %d activation(s) saturated at the f16 limit (65504) and %d were not finite in this video's forwards: the features are unreliable "
                      "(bf16 keeps the range: load_ft_model(...).i3d.compute_dtype = 'bf16' at 3e-3 rel-L2)" % (sat, bad), RuntimeWarning)
    return sat + bad


@torch.no_grad()
def extract_video_sharded(ft_model, clips_local: torch.Tensor, T: int, ncrops: int = 1, batch: int = 75,
                          group=None, fa_model=None, layout: str = "reference", fa_batch: int = 75) -> torch.Tensor:
    """Multi-GPU extraction of ONE video. Each rank passes ITS block of clips
    (sharding.shard_range(T, rank, world) clip times x ncrops crops, crop-minor order); returns the full
    (T, ncrops, F) fp32 tensor on every rank after one RCCL all-gather.

    Without `fa_model` the block is the ft input itself, (T_r*ncrops, 3, 16, H, W). With `fa_model` it is the loader's layout
    (T_r*ncrops, 16, 3, H, W) and every clip goes through the anonymizer first, exactly as the reference's extractors do with their
    hard-coded `anonymized = True` (dali_extraction.py:108,169-178, st_feature_extraction.py:24-30): frames -> fa -> the Q1 reshape
    (`layout='reference'`) or the geometrically meaningful permute -> ft.extract_features; `fa_batch` clips per anonymizer forward."""
    if fa_model is None:
        f = extract_clip_features(ft_model, clips_local, batch)
    else:
        f = extract_anonymized_clip_features(ft_model, fa_model, clips_local, batch, fa_batch, layout)      # an empty shard launches nothing and still joins the collective
    f = f.view(-1, ncrops, f.shape[1])
    return sharding.gather_video_features(f, T, group)


@torch.no_grad()
def extract_video_features_uint8(ft_model, frames: torch.Tensor, cropping_factor: float = 0.8, no_ar_distortion: bool = False, out_hw=(224, 224),
                                 clip_step: int = 32, frame_step: int = 2, t_clip: int = 16, n_clips: int = None, batch: int = 375, streams: int = 2,
                                 out: torch.Tensor = None) -> torch.Tensor:
    """The whole per-video path of dali_extraction.py without its anonymizer, from DECODED FRAMES: frames (T,H,W,3) uint8 (or the float frames DALI hands
    over) resident on the GPU -> HybridValPipe's clip sampling (:62-73: 16 frames, every 2nd, a clip per 32 source frames; frames past the end are zero
    frames) -> val_augmentations (:38-50: /255, centre crop 0.8, antialiased resize) -> I3Res50.extract_features -> (n_clips, F) fp32 rows on the GPU.
    Pre-processing writes the persistent stem's 16-bit input records directly (preprocess.crop_resize_records -> I3Res50.extract_features_records): ONE launch
    per batch in front of the encoder, no fp32 clip batch in HBM. Encoders without the record path (InceptionI3d) take the fp32 clips."""
    from . import engine as E, preprocess
    E.require_cuda(frames, "extract_video_features_uint8")
    net = ft_model.i3d if hasattr(ft_model, "i3d") else ft_model
    t, h, w, c = frames.shape
    if n_clips is None:
        n_clips = -(-t // clip_step)
    if no_ar_distortion:
        ch = cw = int(min(h, w) * cropping_factor)
    else:
        ch, cw = int(h * cropping_factor), int(w * cropping_factor)
    box = preprocess.center_crop_box(h, w, ch, cw)
    if out is None:
        out = torch.empty((n_clips, feature_width(ft_model)), dtype=torch.float32, device=frames.device)
    if n_clips == 0:
        return out
    dev = frames.device
    pool = _STREAMS.setdefault((dev, streams), [torch.cuda.Stream(device=dev) for _ in range(max(1, streams))])
    main = torch.cuda.current_stream(dev)
    for st in pool:
        st.wait_stream(main)
    rec_path = hasattr(net, "extract_features_records") and E.STEM_PT and E.STEM_POOL
    stem = net.packed()["stem_pt"] if rec_path else None
    for j, (i, k) in enumerate(sharding.batch_plan(n_clips, batch, len(pool))):
        with torch.cuda.stream(pool[0 if E.tuning_pending() else j % len(pool)]):
            if rec_path:
                rec = preprocess.crop_resize_records(frames, box, out_hw, stem, k, first=i * clip_step, clip_step=clip_step, frame_step=frame_step, t_clip=t_clip)
                out[i:i + k] = net.extract_features_records(rec).flatten(1)
            else:
                clips = torch.zeros((k, c, t_clip, out_hw[0], out_hw[1]), dtype=torch.float32, device=dev)
                for q in range(k):
                    f0 = (i + q) * clip_step
                    idx = [f0 + f * frame_step for f in range(t_clip) if f0 + f * frame_step < t]
                    if idx:
                        sel = frames[idx[0]:idx[-1] + 1:frame_step].contiguous()
                        preprocess.crop_resize(sel, box, out_hw, out=clips[q, :, :len(idx)], layout="cthw")
                out[i:i + k] = _extract_fn(ft_model)(clips).flatten(1)
    for st in pool:
        main.wait_stream(st)
    return out


def share_tile_choices(ft_model, group=None, src: int = 0) -> int:
    """Multi-GPU jobs: broadcast rank `src`'s decided conv tile configurations (engine.export_tile_table of the packed network) to every rank, so that all
    ranks run the same tiles -- identical clips then give bit-identical features on every GPU and no rank spends its first forwards tuning. Call it once
    after rank `src` has run a few dozen forwards of the batch sizes in use (bench.py does). Returns the number of choices this rank took over."""
    import torch.distributed as dist
    from . import engine as E
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    net = ft_model.i3d if hasattr(ft_model, "i3d") else ft_model
    box = [E.export_tile_table(net.packed()) if dist.get_rank(group) == src else None]
    dist.broadcast_object_list(box, src=src, group=group)
    return E.import_tile_table(net.packed(), box[0]) if dist.get_rank(group) != src else 0


def save_video_features(save_dir: str, video_path: str, feats) -> str:
    """`<video basename without .mp4/.avi>.npy`, float64 C-order (dali_extraction.py:159,182)."""
    name = os.path.basename(video_path)
    for ext in (".mp4", ".avi"):
        name = name.replace(ext, "")
    path = os.path.join(save_dir, name + ".npy")
    arr = feats.detach().cpu().numpy() if torch.is_tensor(feats) else np.asarray(feats)
    if arr.ndim == 3 and arr.shape[1] == 1:
        arr = arr[:, 0]
    np.save(path, np.ascontiguousarray(arr, dtype=np.float64))
    return path


def save_features_batched(save_dir: str, items) -> list:
    """Batched `.npy` writer: items = [(video_path, feats (T,F) or (T,ncrops,F) tensor), ...] of several videos.
    All features cross PCIe in ONE device-to-host copy (one sync instead of one per clip as in
    st_feature_extraction.py:31-37), then each video is written with the reference's naming / float64 layout."""
    items = list(items)
    if not items:
        return []
    flat = torch.cat([f.detach().reshape(-1).to(torch.float32) for _, f in items])
    host = flat.cpu().numpy()
    paths, off = [], 0
    for path, f in items:
        n = f.numel()
        paths.append(save_video_features(save_dir, path, host[off:off + n].reshape(tuple(f.shape))))
        off += n
    return paths
