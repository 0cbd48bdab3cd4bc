"""ORACLE (test infrastructure only) — the reference's 3-D input transforms, restated on torch CPU.

Pinned by tests/golden/pipeline3d.npz (outputs of the reference's own capstone/volumetric/transforms.py, see
tests/golden/make_golden.py::pipeline3d).

* ``resize3d_image`` / ``resize3d_mask``  = ``Resize3D.apply`` / ``apply_to_mask`` (capstone/volumetric/transforms.py:14-22):
  ``F.interpolate(x, size)`` with the default mode "nearest": src = min(int(floorf(dst * (float(in) / out))), in - 1).
* ``to_hwd`` = ``ToTensorV3`` (:39-43): (C,D,H,W) -> (C,H,W,D), masks (D,H,W) -> (H,W,D).
* ``instance`` = what ``MiccaiDataset3D.__getitem__`` (capstone/volumetric/datasets.py:24-48) returns with the transform
  of volumetric/predefined.py:4-7, plus the label map ``_squash_masks_3D`` (volumetric/utils.py:4-7) makes of it.
* ``apply_window`` = capstone/transforms/transforms_2d.py:97-107 (the HU windowing the 2-D path applies; optional here).
"""
import numpy as np
import torch
import torch.nn.functional as F

from .metrics import squash_masks


def nearest_index(out_size, in_size):
    """explicit form of torch's nearest source index (float32 scale, floorf) — used to cross-check F.interpolate"""
    scale = np.float32(in_size) / np.float32(out_size)
    return np.minimum(np.floor(np.arange(out_size, dtype=np.float32) * scale).astype(np.int64), in_size - 1)


def resize3d_image(image, size):
    return F.interpolate(torch.as_tensor(image).unsqueeze(0), tuple(size)).squeeze(0)


def resize3d_mask(mask, size):
    return F.interpolate(torch.as_tensor(mask).unsqueeze(0).unsqueeze(0), tuple(size)).squeeze(0).squeeze(0)


def to_hwd(img):
    return img.permute(0, 2, 3, 1)


def apply_window(image, window_width, window_level, shift=True):
    lo = window_level - (window_width // 2)
    hi = window_level + (window_width // 2)
    clipped = np.clip(image, lo, hi)
    if shift:
        clipped = (clipped - lo) / (hi - lo + 1e-8)
    return clipped


def instance(image, masks, size, window=None):
    """image (1,D,H,W), masks (9,D,H,W) -> image (1,H',W',D') fp32, masks (9,H',W',D') u8, labels (H',W',D') int64"""
    if window is not None:
        image = apply_window(np.asarray(image, dtype=np.float32), *window).astype(np.float32)
    img = to_hwd(resize3d_image(image, size)).contiguous()
    m = torch.stack([resize3d_mask(mk, size).permute(1, 2, 0) for mk in torch.as_tensor(masks)])
    labels = squash_masks(m.unsqueeze(0), m.shape[0] + 1)[0]
    return img, m, labels
