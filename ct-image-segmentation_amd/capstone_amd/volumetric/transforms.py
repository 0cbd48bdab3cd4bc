"""Device-side 3-D input pipeline — drop-in for reference capstone/volumetric/transforms.py.

``Resize3D`` (:9-32, ``F.interpolate`` nearest to ``size`` = D×H×W) and ``ToTensorV3`` (:35-49, (C,D,H,W) -> (C,H,W,D)) keep
their names, constructor arguments and ``apply`` / ``apply_to_mask`` methods but take **GPU tensors** (the raw ``.npz`` arrays
uploaded once) and run as one HIP pass, ``ctseg_resize3d_to_hwd``: the resize writes (H,W,D)-contiguous storage straight away and
``Resize3D`` hands back the (D,H,W) *view* of it, so ``ToTensorV3``'s permute — the composition of
volumetric/predefined.py:4-7 — ends on a contiguous tensor with no further copy.

``InstancePipeline3D`` is the fused form used by ``datasets.MiccaiDataset3D``: image resize + permute (+ optional HU window,
capstone/transforms/transforms_2d.py:97-107, which the reference's 3-D path omits) and the nine masks resized, permuted and —
with ``squash=True`` — reduced to the label map of ``_squash_masks_3D`` (volumetric/utils.py:4-7) in the same pass: 1 byte
per voxel leaves the kernel instead of 9, and the trainer's squash pass disappears.
There is no CPU fallback: CPU tensors raise ``NativeError``.
"""
from typing import Optional, Sequence, Tuple

import torch

from .. import _native as nat

_IMG_DTYPES = {torch.float32: nat.F32, torch.int16: nat.I16, torch.uint8: nat.U8}


def _image_code(image):
    if image.dtype not in _IMG_DTYPES:
        image = image.float()
    return image.contiguous(), _IMG_DTYPES[image.dtype]


def _window_args(window):
    """window = None | (width, level) | (width, level, shift): the reference's apply_window(image, width, level, shift)"""
    if window is None:
        return 0, 0.0, 1.0
    width, level = window[0], window[1]
    shift = window[2] if len(window) > 2 else True
    return (2 if shift else 1), float(level - (width // 2)), float(level + (width // 2))


def resize3d_to_hwd(image: Optional[torch.Tensor], masks: Optional[torch.Tensor], size: Sequence[int], window=None,
                    want_masks: bool = True, want_labels: bool = False):
    """image (D,H,W) f32/i16/u8, masks (K,D,H,W) u8 -> image (H',W',D') fp32, masks (K,H',W',D') u8, labels (H',W',D') u8, hist (K+1)"""
    ref = image if image is not None else masks
    nat.require_gpu(ref, "Resize3D")
    D, H, W = ref.shape[-3:]
    Do, Ho, Wo = (int(v) for v in size)
    dev = ref.device
    img_out = m_out = lab = hist = None
    code, K = nat.F32, 0
    if image is not None:
        image, code = _image_code(image.reshape(D, H, W))
        img_out = torch.empty((Ho, Wo, Do), dtype=torch.float32, device=dev)
    if masks is not None:
        masks = masks.reshape(-1, D, H, W)
        if masks.dtype != torch.uint8:
            masks = masks.to(torch.uint8)
        masks = masks.contiguous()
        K = masks.shape[0]
        if want_masks:
            m_out = torch.empty((K, Ho, Wo, Do), dtype=torch.uint8, device=dev)
        if want_labels:
            lab = torch.empty((Ho, Wo, Do), dtype=torch.uint8, device=dev)
            hist = torch.zeros(K + 1, dtype=torch.int64, device=dev)
    wmode, lo, hi = _window_args(window)
    nat.call("ctseg_resize3d_to_hwd", nat.ptr(image), code, nat.ptr(masks), K, D, H, W, Do, Ho, Wo, wmode, lo, hi,
             nat.ptr(img_out), nat.ptr(m_out), nat.ptr(lab), nat.ptr(hist))
    return img_out, m_out, lab, hist


class Resize3D:
    def __init__(self, size=(96, 256, 256), always_apply=True, p=1.0):
        self.size = tuple(size)      # DxHxW, as the reference

    def apply(self, image: torch.Tensor, **params) -> torch.Tensor:
        """(1,D,H,W) -> (1,D',H',W') (values of F.interpolate(image[None], size)[0]; storage is (H',W',D')-contiguous)"""
        out = resize3d_to_hwd(image, None, self.size)[0]
        return out.permute(2, 0, 1).unsqueeze(0)

    def apply_to_mask(self, image: torch.Tensor, **params) -> torch.Tensor:
        """(D,H,W) -> (D',H',W')"""
        out = resize3d_to_hwd(None, image.unsqueeze(0), self.size)[1]
        return out[0].permute(2, 0, 1)

    def get_transform_init_args_names(self):
        return []


class ToTensorV3:
    def __init__(self, always_apply=True, p=1.0):
        pass

    def apply(self, img, **params):
        return img.permute(0, 2, 3, 1)

    def apply_to_mask(self, mask, **params):
        return mask.permute(1, 2, 0)

    def get_transform_init_args_names(self):
        return ("transpose_mask",)


class InstancePipeline3D:
    """``transform(image=..., masks=...) -> {"image": (1,H,W,D) fp32, "masks": ...}`` like the albumentations Compose of
    volumetric/predefined.py:4-7.  ``squash=False``: "masks" = (9,H,W,D) uint8 (the reference's batch element);
    ``squash=True``: "masks" = (H,W,D) uint8 label map with ``"hist"`` = per-class voxel counts riding along."""

    def __init__(self, size=(96, 256, 256), squash: bool = False, window: Optional[Tuple] = None):
        self.size, self.squash, self.window = tuple(size), squash, window

    def __call__(self, image: torch.Tensor, masks) -> dict:
        if isinstance(masks, (list, tuple)):
            masks = torch.stack(list(masks))
        img, m, lab, hist = resize3d_to_hwd(image, masks, self.size, self.window, want_masks=not self.squash,
                                            want_labels=self.squash)
        out = {"image": img.unsqueeze(0), "masks": lab if self.squash else m}
        if self.squash:
            out["hist"] = hist
        return out


windowed_degree_0 = {"train": InstancePipeline3D(), "test": InstancePipeline3D()}      # volumetric/predefined.py:4-7
