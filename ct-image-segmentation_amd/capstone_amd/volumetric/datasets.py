"""Device-fed counterpart of reference capstone/volumetric/datasets.py.

``MiccaiDataset3D`` reads the same ``.npz`` instances (``image`` 1xDxHxW, ``masks`` 9xDxHxW, ``mask_indicator`` 9 — written by
capstone/data/process_miccai.py:96-131), uploads the raw arrays once and runs the transform on the MI355X
(``transforms.InstancePipeline3D``) instead of albumentations on the host.  ``__getitem__`` returns the reference's triple
``(image (1,H,W,D), masks (9,H,W,D), mask_indicator (9,))`` as device tensors, or — with a ``squash=True`` pipeline —
``masks`` already reduced to the (H,W,D) uint8 label map; ``collate_3d`` stacks either form into the batch ``BaseUNet3D``
takes (pre-squashed label maps are recognised by ``_squash_masks_3D`` / ``fit_step`` and skip the squash pass).
"""
from pathlib import Path

import numpy as np
import torch

from .. import STRUCTURES
from .transforms import InstancePipeline3D


class MiccaiDataset3D:
    def __init__(self, path: str, transform=None, device="cuda"):
        self.path = Path(path).absolute()
        self.transform = transform
        self.device = torch.device(device)
        self.instance_paths = sorted(p.as_posix() for p in self.path.iterdir())      # same order on every platform (ref :16-18)

    def __len__(self) -> int:
        return len(self.instance_paths)

    def __getitem__(self, index: int):
        instance = np.load(self.instance_paths[index])
        image, masks, mask_indicator = instance["image"], instance["masks"], instance["mask_indicator"]
        assert len(mask_indicator) == len(STRUCTURES)
        assert masks.shape[0] == len(STRUCTURES)
        image = torch.from_numpy(np.ascontiguousarray(image)).to(self.device, non_blocking=True)
        masks = torch.from_numpy(np.ascontiguousarray(masks).view(np.uint8) if masks.dtype == np.bool_ else np.ascontiguousarray(masks))
        masks = masks.to(self.device, non_blocking=True)
        hist = None
        if self.transform is not None:
            transformed = self.transform(image=image, masks=masks)
            image, masks, hist = transformed["image"], transformed["masks"], transformed.get("hist")
        mask_indicator = torch.from_numpy(mask_indicator).to(self.device)
        if hist is not None:
            masks._ctseg_hist = hist
        return image, masks, mask_indicator


def collate_3d(samples):
    """list of ``__getitem__`` triples -> (images (B,1,H,W,D), masks, mask_indicator (B,9)); masks = (B,9,H,W,D) uint8, or the
    (B,H,W,D) uint8 label maps of a squashing pipeline carrying ``_ctseg_labels`` = (labels (B,S), per-class counts (B,10))"""
    images = torch.stack([s[0] for s in samples])
    indicator = torch.stack([s[2] for s in samples])
    masks = torch.stack([s[1] for s in samples])
    if all(hasattr(s[1], "_ctseg_hist") for s in samples):
        masks._ctseg_labels = (masks.reshape(len(samples), -1), torch.stack([s[1]._ctseg_hist for s in samples]))
    return images, masks, indicator


def get_miccai_3d(split: str = "train", transform=None, root: str = "storage", device="cuda"):
    assert split in ["train", "valid", "test"], "Invalid data split passed"
    if transform is None:
        transform = InstancePipeline3D()
    return MiccaiDataset3D(f"{root}/miccai_3d/{split}", transform=transform, device=device)
