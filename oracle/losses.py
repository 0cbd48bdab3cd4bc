"""Oracle: segmentation losses of the reference's 3-D training step.

TEST INFRASTRUCTURE (see oracle/__init__.py).

Reference anchors:
  CLASS_WEIGHT                capstone/models/losses.py:10-21 (= volumetric/losses.py:10-21)
  cross_entropy / weighted    capstone/models/losses.py:45-68 (F.cross_entropy on (B,C,*sp) logits)
  generalized_dice_loss       capstone/models/temp.py:96-170 (w=1/sum(y)^2, inf -> per-sample max)
  missing_mask                capstone/models/losses.py:206-221
  MultipleLoss                capstone/models/losses.py:170-203, volumetric/losses.py:119-130
  dice_loss, focal_loss       monai==0.3 DiceLoss / FocalLoss as configured at
                              capstone/volumetric/losses.py:72-77,107 — third-party, absent
                              here: **parity unpinned** (formulas per SURVEY.md §8 a9').

Deliberate fix, flagged: the reference's ``MultipleLossWrapper3D`` resolves names in the
*2-D* registry, whose Dice/Focal wrappers assert 4-D input and therefore raise on volumes
(SURVEY.md §3.1 bug 1). The oracle wires the 3-D registry the file evidently intended
(volumetric/losses.py:119-125), i.e. the same maths without the ndim asserts.
"""
import torch
import torch.nn.functional as F

from .metrics import N_CLASSES, one_hot

CLASS_WEIGHT = (1e-10, 0.007, 0.3296, 0.0046, 0.2619, 0.3035, 0.0068, 0.0065, 0.0374, 0.0426)
SMOOTH = 1e-5


def cross_entropy(logits, target):
    return F.cross_entropy(logits, target)


def weighted_cross_entropy(logits, target):
    return F.cross_entropy(logits, target, weight=torch.tensor(CLASS_WEIGHT).type_as(logits))


def _fg_probs_and_truth(logits, target):
    c = logits.shape[1]
    p = torch.softmax(logits, 1)[:, 1:]
    y = one_hot(target.unsqueeze(1), c)[:, 1:]
    return p, y, list(range(2, logits.ndim))


def _reduce(f, reduction):
    return f.mean() if reduction == "mean" else f


def dice_loss(logits, target, reduction="mean"):
    """DiceLoss(include_background=False, to_onehot_y=True, softmax=True): (B,9) or scalar."""
    p, y, ax = _fg_probs_and_truth(logits, target)
    inter = (p * y).sum(ax)
    denom = y.sum(ax) + p.sum(ax)
    return _reduce(1.0 - (2.0 * inter + SMOOTH) / (denom + SMOOTH), reduction)


def generalized_dice_loss(logits, target, reduction="mean"):
    p, y, ax = _fg_probs_and_truth(logits, target)
    inter = (p * y).sum(ax)
    g = y.sum(ax)
    denom = g + p.sum(ax)
    w = torch.reciprocal(g.float() * g.float())
    for row in w:  # per sample: infinite weights (empty class) -> the sample's largest finite weight
        inf = torch.isinf(row)
        row[inf] = 0.0
        row[inf] = torch.max(row)
    return _reduce(1.0 - (2.0 * (inter * w) + SMOOTH) / (denom * w + SMOOTH), reduction)


def focal_loss(logits, target, reduction="mean", gamma=2.0):
    """FocalLoss(gamma=2) on a one-hot target: per (b,c) voxel-mean of -(1-p)^g * t * log p."""
    b, c = logits.shape[:2]
    t = one_hot(target.unsqueeze(1), c).reshape(b, c, -1)
    logp = F.log_softmax(logits.reshape(b, c, -1), dim=1)
    w = torch.pow(1.0 - torch.exp(logp), gamma)
    return _reduce(torch.mean(-w * t * logp, dim=-1), reduction)


def missing_mask(name, loss, indicator, n_classes=N_CLASSES):
    """AnatomyNet-style weighting of a (B,C) loss table by annotation availability."""
    if name == "Focal":
        bg = (indicator.sum(dim=1, keepdim=True) == (n_classes - 1)).float()
        indicator = torch.cat([bg, indicator], dim=1)
    w = 1.0 / indicator.sum(dim=0)
    if torch.any(torch.isinf(w)):
        w = torch.ones_like(w)
    w = w / w.sum()
    return (loss * w[None, :] * indicator).sum(dim=1).mean()


_TABLE = {
    "CrossEntropy": lambda x, t, r: cross_entropy(x, t),
    "WeightedCrossEntropy": lambda x, t, r: weighted_cross_entropy(x, t),
    "Dice": dice_loss,
    "GeneralizedDice": generalized_dice_loss,
    "Focal": focal_loss,
}


class MultipleLoss:
    def __init__(self, losses, exclude_missing=False):
        for name in losses:
            assert name in _TABLE
        self.names, self.exclude_missing = list(losses), exclude_missing

    def __call__(self, input, target, mask_indicator=None):
        if mask_indicator is not None:
            mask_indicator = mask_indicator.type_as(input)
        red = "none" if self.exclude_missing else "mean"
        out = {}
        for name in self.names:
            v = _TABLE[name](input, target, red)
            if self.exclude_missing and name not in ("CrossEntropy", "WeightedCrossEntropy"):
                v = missing_mask(name, v, mask_indicator)
            out[name] = v
        return out
