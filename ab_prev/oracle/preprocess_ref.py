"""TEST INFRASTRUCTURE ONLY (CPU oracle; imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg,
never by the product path).

Restatement of `DALIDataloader.val_augmentations` (feature_extraction/dali_extraction.py:38-50).

Pinning: torchvision (==0.15.2, pip_requirements.txt:78) is NOT installed in this image, so the two torchvision calls
are restated from that version's published source:
  * F.center_crop(tensor, (ch,cw)): top = int(round((h-ch)/2.0)), left = int(round((w-cw)/2.0))   -- PARITY UNPINNED
    (integer box arithmetic only)
  * F.resize(tensor, (h,w), antialias=True) on a float tensor: torch.nn.functional.interpolate(img, size,
    mode='bilinear', align_corners=False, antialias=True) -- the arithmetic itself is torch's, imported directly here.
"""
import torch
import torch.nn.functional as F


def center_crop(video, ch, cw):
    h, w = video.shape[-2:]
    top, left = int(round((h - ch) / 2.0)), int(round((w - cw) / 2.0))
    return video[..., top:top + ch, left:left + cw]


def val_augmentations(video, cropping_factor=0.8, no_ar_distortion=False, reso_h=224, reso_w=224):
    """video (1,T,H,W,C) float 0..255 -> (1,T,C,reso_h,reso_w); line-by-line dali_extraction.py:38-50."""
    video = torch.transpose(video, 2, 4)
    video = torch.transpose(video, 3, 4)
    video = video / 255.
    ori_w, ori_h = int(video.shape[-1]), int(video.shape[-2])
    m = min(ori_h, ori_w)
    if no_ar_distortion:
        video = center_crop(video.squeeze(), int(m * cropping_factor), int(m * cropping_factor))
    else:
        video = center_crop(video.squeeze(), int(ori_h * cropping_factor), int(ori_w * cropping_factor))
    video = F.interpolate(video, size=(reso_h, reso_w), mode="bilinear", align_corners=False, antialias=True)
    return video.unsqueeze(dim=0)


def resize_matrix(in_size, out_size):
    """Dense (out,in) matrix of the 1-D antialiased bilinear resize, read off torch itself (identity columns)."""
    # N=1, C=in (one channel per basis vector), H=in, W=2 (a width-1 image takes a degenerate path in torch 2.10)
    eye = torch.eye(in_size).view(1, in_size, in_size, 1).expand(1, in_size, in_size, 2).contiguous()
    return F.interpolate(eye, size=(out_size, 2), mode="bilinear", align_corners=False, antialias=True)[0, :, :, 0].t()


def shanghai_augmentation(image, cropping_factor=0.8, no_ar_distortion=False, reso_h=224, reso_w=224):
    """`shanghai_frames_dataset.augmentation` (feature_extraction/shanghai_dl.py:27-40) for one (H,W,3) uint8 numpy frame ->
    (3,reso_h,reso_w) fp32. torchvision's PIL branch restated (to_pil_image = Image.fromarray; center_crop = the box rule above +
    Image.crop; resize = Image.resize(size[::-1], BILINEAR); to_tensor = uint8 HWC -> CHW float / 255); the resampling itself is
    PILLOW'S OWN (installed in this image): pinned. The crop arithmetic follows the reference lines literally, including
    `min(image.shape)` (minimum over H, W and the channel count) and the square crop of side int(H * factor)."""
    import numpy as np
    from PIL import Image
    ori_h, ori_w = image.shape[0], image.shape[1]
    min_size = min(image.shape)
    pil = Image.fromarray(image)
    side = int(min_size * cropping_factor) if no_ar_distortion else int(ori_h * cropping_factor)
    top, left = int(round((ori_h - side) / 2.0)), int(round((ori_w - side) / 2.0))
    pil = pil.crop((left, top, left + side, top + side))
    pil = pil.resize((reso_w, reso_h), Image.BILINEAR)
    arr = np.array(pil, copy=True)
    return torch.from_numpy(arr).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
