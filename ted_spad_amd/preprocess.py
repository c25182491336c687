"""GPU-side frame pre-processing that feeds the encoder: the build's counterpart of
`DALIDataloader.val_augmentations` (feature_extraction/dali_extraction.py:38-50):

    (1,T,H,W,C) decoder frames -> transpose -> /255 -> F.center_crop(factor 0.8) -> F.resize((224,224), antialias=True)

done by ONE HIP kernel per crop box (tedspad_frames_crop_resize: divide, crop, antialiased separable resize through
LDS, optional flip, strided fp32 store), so a clip can be written straight into the (n,3,16,h,w) batch the encoder
takes. torchvision (0.15.2, pip_requirements.txt:78) is not installed here: `center_crop`'s box arithmetic is
restated from its published source (parity unpinned for that integer rounding rule); the resize arithmetic is the
`torch.nn.functional.interpolate(mode='bilinear', antialias=True)` call torchvision makes for float tensors, which IS
importable and pins oracle/preprocess_ref.py.

`shanghai_frames_dataset.augmentation` (shanghai_dl.py:27-40) goes through PIL images (uint8 intermediate rounding
inside PIL's resize); that variant is not built.
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
