"""GPU-side frame pre-processing that feeds the encoder: the build's counterpart of
`DALIDataloader.val_augmentations` (feature_extraction/dali_extraction.py:38-50):

    (1,T,H,W,C) decoder frames -> transpose -> /255 -> F.center_crop(factor 0.8) -> F.resize((224,224), antialias=True)

done by ONE HIP kernel per crop box (tedspad_frames_crop_resize: divide, crop, antialiased separable resize through
LDS, optional flip, strided fp32 store), so a clip can be written straight into the (n,3,16,h,w) batch the encoder
takes. torchvision (0.15.2, pip_requirements.txt:78) is not installed here: `center_crop`'s box arithmetic is
restated from its published source (parity unpinned for that integer rounding rule); the resize arithmetic is the
`torch.nn.functional.interpolate(mode='bilinear', antialias=True)` call torchvision makes for float tensors, which IS
importable and pins oracle/preprocess_ref.py.

`shanghai_frames_dataset.augmentation` (shanghai_dl.py:27-40) goes through PIL images: torchvision's `resize` of a PIL image is
Pillow's `Image.resize(BILINEAR)`, a two-pass fixed-point resample with a uint8 intermediate image. `shanghai_augmentation` below
reproduces it bit for bit on the GPU (tedspad_frames_crop_resize_pil); Pillow IS installed in this image, so oracle/preprocess_ref.py
calls it directly (pinned), only torchvision's `center_crop` box rule stays restated.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check
from .engine import _stream_ptr, require_cuda

_TABLES = {}


def aa_table_host(in_size: int, out_size: int) -> np.ndarray:
    """(out_size, 2 + taps) int32 words {first index, count, float weights...} from the library's host builder."""
    taps = _lib.lib().tedspad_resize_aa_taps(int(in_size), int(out_size))
    tab = np.zeros((out_size, 2 + taps), dtype=np.int32)
    check(_lib.lib().tedspad_resize_aa_table(int(in_size), int(out_size), tab.ctypes.data), "tedspad_resize_aa_table")
    return tab


def _table(in_size, out_size, device):
    key = (int(in_size), int(out_size), str(device))
    t = _TABLES.get(key)
    if t is None:
        t = torch.from_numpy(aa_table_host(in_size, out_size)).to(device)
        _TABLES[key] = t
    return t


def center_crop_box(h: int, w: int, ch: int, cw: int):
    """torchvision.transforms.functional.center_crop's box: top = int(round((h - ch) / 2.0)) (Python round:
    half to even), same for left. Crops larger than the frame (which torchvision zero-pads) are not supported."""
    if ch > h or cw > w:
        raise ValueError("center_crop_box: crop %dx%d larger than the %dx%d frame" % (ch, cw, h, w))
    return int(round((h - ch) / 2.0)), int(round((w - cw) / 2.0)), ch, cw


def ten_crop_boxes(h: int, w: int, ch: int, cw: int):
    """torchvision ten_crop order (tl, tr, bl, br, center, then the same five of the horizontally flipped frame) as
    (y0, x0, ch, cw, flip) boxes in the coordinates of the UNFLIPPED frame. Not used by the reference's extractor
    (single centre crop); provided for the (T,10,F) layout its MGFN loader accepts (dataset.py:70-89)."""
    five = [(0, 0), (0, w - cw), (h - ch, 0), (h - ch, w - cw), center_crop_box(h, w, ch, cw)[:2]]
    out = [(y, x, ch, cw, False) for y, x in five]
    out += [(y, w - cw - x, ch, cw, True) for y, x in five]
    return out


def crop_resize(frames: torch.Tensor, box, out_hw, flip: bool = False, out: torch.Tensor = None, layout: str = "tchw",
                divisor: float = 255.0) -> torch.Tensor:
    """frames: (T,H,W,C) uint8 or float32 on the GPU, contiguous. Returns / fills fp32 `out`:
    layout 'tchw' -> (T,C,oh,ow) (what val_augmentations returns), 'cthw' -> (C,T,oh,ow) (one encoder clip).
    `out` may be any strided view of the right shape (e.g. batch[i] of a (n,3,16,h,w) clip batch)."""
    require_cuda(frames, "crop_resize")
    if frames.dim() != 4 or frames.dtype not in (torch.uint8, torch.float32) or not frames.is_contiguous():
        raise ValueError("crop_resize: frames must be a contiguous (T,H,W,C) uint8/float32 tensor")
    t, h, w, c = frames.shape
    y0, x0, ch, cw = [int(v) for v in box]
    oh, ow = int(out_hw[0]), int(out_hw[1])
    shape = (t, c, oh, ow) if layout == "tchw" else (c, t, oh, ow)
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=frames.device)
    if tuple(out.shape) != shape or out.dtype != torch.float32:
        raise ValueError("crop_resize: out must be fp32 %s" % (shape,))
    s = out.stride()
    so = (s[0], s[1]) if layout == "tchw" else (s[1], s[0])
    ytab, xtab = _table(ch, oh, frames.device), _table(cw, ow, frames.device)
    check(_lib.lib().tedspad_frames_crop_resize(frames.data_ptr(), int(frames.dtype == torch.float32), t, h, w, c, y0, x0, ch, cw, oh, ow,
                                                ytab.data_ptr(), xtab.data_ptr(), C.c_float(divisor), int(flip), out.data_ptr(),
                                                so[0], so[1], s[2], s[3], _stream_ptr()), "tedspad_frames_crop_resize")
    return out


def crop_resize_records(frames: torch.Tensor, box, out_hw, stem, n_clips: int, first: int = 0, clip_step: int = 32, frame_step: int = 2,
                        t_clip: int = 16, flip: bool = False, out: torch.Tensor = None, divisor: float = 255.0) -> torch.Tensor:
    """frames: (T,H,W,C) uint8 or float32 decoder frames on the GPU -> the persistent stem's 16-bit input records X[n][tp][oh][2][ow/2][24]
    (engine.StemPT.layout's tensor) of `n_clips` clips, clip i = frames first + i*clip_step + f*frame_step (HybridValPipe's sampling,
    dali_extraction.py:62-73), each frame / divisor -> crop `box` -> antialiased resize to `out_hw` (val_augmentations, :38-50). ONE launch; the fp32 clip
    batch the encoder's public entry takes is never materialised. `stem`: the engine.StemPT of the encoder (I3Res50.packed()["stem_pt"]): temporal
    geometry and storage type. Feed the result to I3Res50.extract_features_records."""
    require_cuda(frames, "crop_resize_records")
    if frames.dim() != 4 or frames.dtype not in (torch.uint8, torch.float32) or not frames.is_contiguous():
        raise ValueError("crop_resize_records: frames must be a contiguous (T,H,W,C) uint8/float32 tensor")
    t, h, w, c = frames.shape
    y0, x0, ch, cw = [int(v) for v in box]
    oh, ow = int(out_hw[0]), int(out_hw[1])
    tp = stem.frame_pairs(t_clip)
    shape = (n_clips, tp, oh, 2, ow // 2, 24)
    if out is None:
        out = torch.empty(shape, dtype=stem.torch_dtype, device=frames.device)
    if tuple(out.shape) != shape or out.dtype != stem.torch_dtype or not out.is_contiguous():
        raise ValueError("crop_resize_records: out must be a contiguous %s tensor of %s" % (shape, stem.torch_dtype))
    ytab, xtab = _table(ch, oh, frames.device), _table(cw, ow, frames.device)
    check(_lib.lib().tedspad_frames_crop_resize_tp(frames.data_ptr(), int(frames.dtype == torch.float32), t, h, w, c, int(n_clips), int(first), int(clip_step),
                                                   int(frame_step), int(t_clip), y0, x0, ch, cw, oh, ow, ytab.data_ptr(), xtab.data_ptr(), C.c_float(divisor),
                                                   int(flip), out.data_ptr(), stem.pad_t, stem.stride_t, tp, stem.dtype_code, _stream_ptr()),
          "tedspad_frames_crop_resize_tp")
    return out


def val_augmentations(video: torch.Tensor, cropping_factor: float = 0.8, no_ar_distortion: bool = False, reso_h: int = 224,
                      reso_w: int = 224) -> torch.Tensor:
    """dali_extraction.py:38-50. video: (1,T,H,W,C) frames with values 0..255 (uint8, or float as DALI delivers them)
    -> (1,T,C,reso_h,reso_w) fp32 in [0,1]."""
    if video.dim() != 5 or video.shape[0] != 1:
        raise ValueError("val_augmentations: expected (1,T,H,W,C) like the DALI reader (batch size 1)")
    _, t, h, w, c = video.shape
    if no_ar_distortion:
        m = min(h, w)
        ch = cw = int(m * cropping_factor)
    else:
        ch, cw = int(h * cropping_factor), int(w * cropping_factor)
    box = center_crop_box(h, w, ch, cw)
    return crop_resize(video[0].contiguous(), box, (reso_h, reso_w)).unsqueeze(0)


# ---- shanghai_frames_dataset.augmentation (shanghai_dl.py:27-40): the PIL path ----------------------------------------------------

_PIL_TABLES = {}
PIL_PRECISION_BITS = 32 - 8 - 2          # libImaging/Resample.c


def pil_table(in_size: int, out_size: int):
    """Pillow's BILINEAR resample coefficients for in_size -> out_size (no box offset), as libImaging/Resample.c computes them:
    precompute_coeffs (float64: support = max(scale, 1), triangle filter, normalised) then normalize_coeffs_8bpc
    (kk = trunc(+-0.5 + k * 2^22)). Returns (int32 (out_size, 2 + ksize) array {xmin, count, kk...}, ksize)."""
    scale = float(in_size) / float(out_size)
    filterscale = scale if scale >= 1.0 else 1.0
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    tab = np.zeros((out_size, 2 + ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = np.zeros(xmax, dtype=np.float64)
        ww = 0.0
        for x in range(xmax):
            a = (x + xmin - center + 0.5) * ss
            if a < 0.0:
                a = -a
            w = 1.0 - a if a < 1.0 else 0.0
            k[x] = w
            ww += w
        if ww != 0.0:
            for x in range(xmax):
                k[x] /= ww
        tab[xx, 0], tab[xx, 1] = xmin, xmax
        for x in range(xmax):
            v = k[x] * (1 << PIL_PRECISION_BITS)
            tab[xx, 2 + x] = int(-0.5 + v) if k[x] < 0 else int(0.5 + v)
    return tab, ksize


def _pil_table(in_size, out_size, device):
    key = (int(in_size), int(out_size), str(device))
    t = _PIL_TABLES.get(key)
    if t is None:
        tab, ks = pil_table(in_size, out_size)
        t = (torch.from_numpy(tab).to(device), ks)
        _PIL_TABLES[key] = t
    return t


def crop_resize_pil(frames: torch.Tensor, box, out_hw, out: torch.Tensor = None, layout: str = "tchw") -> torch.Tensor:
    """frames: (T,H,W,C) uint8 on the GPU -> fp32 (T,C,oh,ow) ('tchw') or (C,T,oh,ow) ('cthw'): crop + Pillow BILINEAR resize + /255."""
    require_cuda(frames, "crop_resize_pil")
    if frames.dim() != 4 or frames.dtype != torch.uint8 or not frames.is_contiguous():
        raise ValueError("crop_resize_pil: frames must be a contiguous (T,H,W,C) uint8 tensor (PIL images are 8-bit)")
    t, h, w, c = frames.shape
    y0, x0, ch, cw = [int(v) for v in box]
    oh, ow = int(out_hw[0]), int(out_hw[1])
    shape = (t, c, oh, ow) if layout == "tchw" else (c, t, oh, ow)
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=frames.device)
    if tuple(out.shape) != shape or out.dtype != torch.float32:
        raise ValueError("crop_resize_pil: out must be fp32 %s" % (shape,))
    s = out.stride()
    so = (s[0], s[1]) if layout == "tchw" else (s[1], s[0])
    (ytab, ky), (xtab, kx) = _pil_table(ch, oh, frames.device), _pil_table(cw, ow, frames.device)
    check(_lib.lib().tedspad_frames_crop_resize_pil(frames.data_ptr(), t, h, w, c, y0, x0, ch, cw, oh, ow, ytab.data_ptr(), ky, xtab.data_ptr(), kx,
                                                    out.data_ptr(), so[0], so[1], s[2], s[3], _stream_ptr()), "tedspad_frames_crop_resize_pil")
    return out


def shanghai_crop_size(h: int, w: int, c: int, cropping_factor: float = 0.8, no_ar_distortion: bool = False):
    """The crop `shanghai_frames_dataset.augmentation` takes from an (h, w, c) frame (shanghai_dl.py:28-35), quirks included:
    with no_ar_distortion the reference uses min(image.shape) -- the minimum over (H, W, C), i.e. the channel count 3 -- and
    otherwise a SQUARE of side int(H * factor) (both sides from the height). Reproduced as written."""
    if no_ar_distortion:
        side = int(min(h, w, c) * cropping_factor)
    else:
        side = int(h * cropping_factor)
    return side, side


def shanghai_augmentation(frames: torch.Tensor, cropping_factor: float = 0.8, no_ar_distortion: bool = False, reso_h: int = 224,
                          reso_w: int = 224, out: torch.Tensor = None, layout: str = "tchw") -> torch.Tensor:
    """shanghai_dl.py:27-40 for a stack of decoded frames: (T,H,W,3) uint8 (or one (H,W,3) frame) -> fp32 (T,3,reso_h,reso_w) in [0,1]
    (one frame: (3,reso_h,reso_w)), bit-identical to to_pil_image -> center_crop -> resize(antialias=True) -> to_tensor."""
    single = frames.dim() == 3
    if single:
        frames = frames.unsqueeze(0)
    t, h, w, c = frames.shape
    ch, cw = shanghai_crop_size(h, w, c, cropping_factor, no_ar_distortion)
    if ch < 1 or cw < 1:
        raise ValueError("shanghai_augmentation: empty crop %dx%d" % (ch, cw))
    y = crop_resize_pil(frames.contiguous(), center_crop_box(h, w, ch, cw), (reso_h, reso_w), out=out, layout=layout)
    return y[0] if single and layout == "tchw" else y
