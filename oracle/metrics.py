"""Oracle: label-map squashing, softmax->argmax, mean-Dice metric and its reduction.

TEST INFRASTRUCTURE (see oracle/__init__.py). Pinned by tests/golden/metrics_*.npz,
which were produced by running the reference's own files.

Reference anchors:
  squash_masks          capstone/volumetric/utils.py:4-7, capstone/training/utils.py:13-16
  squash_predictions    capstone/training/utils.py:19-20
  one_hot               monai.networks.one_hot as used at capstone/models/temp.py:131
                        and (via AsDiscrete) capstone/volumetric/metrics.py:19-20
  meandice              capstone/models/temp.py:173-214 (+ ignore_background :217-230)
  metric_reduction      capstone/models/temp.py:233-292
  DiceMetric            capstone/models/metrics.py:8-21, capstone/volumetric/metrics.py:5-20
"""
import torch

N_CLASSES = 10  # 9 structures (capstone/utils/miccai.py:14-24) + background


def squash_masks(masks: torch.Tensor, n_classes: int = N_CLASSES) -> torch.Tensor:
    """(B, n_classes-1, *sp) binary -> (B, *sp) int64 label map; highest set class index wins."""
    ids = torch.arange(1, n_classes, device=masks.device)
    shape = [1, -1] + [1] * (masks.ndim - 2)
    return (masks * ids.view(shape)).max(dim=1).values


def squash_predictions(logits: torch.Tensor) -> torch.Tensor:
    """softmax over channels THEN argmax (first maximal index) — not argmax(logits)."""
    return torch.softmax(logits, dim=1).argmax(dim=1)


def one_hot(labels: torch.Tensor, num_classes: int) -> torch.Tensor:
    """(B,1,*sp) integer labels -> (B,C,*sp) float32 one-hot via scatter."""
    shape = list(labels.shape)
    shape[1] = num_classes
    out = torch.zeros(shape, dtype=torch.float32, device=labels.device)
    return out.scatter_(1, labels.long(), 1.0)


def meandice(y_pred: torch.Tensor, y: torch.Tensor, include_background: bool = True) -> torch.Tensor:
    """Per-(sample, class) Dice of two binarised one-hot tensors; NaN where the truth is empty."""
    if not include_background:
        if y.shape[1] > 1:
            y = y[:, 1:]
        if y_pred.shape[1] > 1:
            y_pred = y_pred[:, 1:]
    y, y_pred = y.float(), y_pred.float()
    if y.shape != y_pred.shape:
        raise ValueError("y_pred and y should have same shapes.")
    axes = list(range(2, y.ndim))
    inter = (y * y_pred).sum(dim=axes)
    t = y.sum(dim=axes)
    denom = t + y_pred.sum(dim=axes)
    nan = torch.tensor(float("nan"), device=y.device)
    return torch.where(t > 0, 2.0 * inter / denom, nan)


_MODES = ("mean", "sum", "mean_batch", "sum_batch", "mean_channel", "sum_channel", "none")


def metric_reduction(f: torch.Tensor, reduction: str = "mean"):
    """NaN-aware reduction of a (batch, class) score table. Returns (value, not_nans).

    Like the reference it zeroes the NaNs of ``f`` IN PLACE (temp.py:252-254).
    """
    if reduction not in _MODES:
        raise ValueError(f"Unsupported reduction: {reduction}")
    nans = torch.isnan(f)
    ok = (~nans).float()
    f[nans] = 0
    zero = torch.zeros(1, dtype=torch.float, device=f.device)
    if reduction == "mean":
        ok = ok.sum(dim=1)
        f = torch.where(ok > 0, f.sum(dim=1) / ok, zero)
        ok = (ok > 0).float().sum(dim=0)
        f = torch.where(ok > 0, f.sum(dim=0) / ok, zero)
    elif reduction == "sum":
        ok = ok.sum(dim=[0, 1])
        f = f.sum(dim=[0, 1])
    elif reduction == "mean_batch":
        ok = ok.sum(dim=0)
        f = torch.where(ok > 0, f.sum(dim=0) / ok, zero)
    elif reduction == "sum_batch":
        ok = ok.sum(dim=0)
        f = f.sum(dim=0)
    elif reduction == "mean_channel":
        ok = ok.sum(dim=1)
        f = torch.where(ok > 0, f.sum(dim=1) / ok, zero)
    elif reduction == "sum_channel":
        ok = ok.sum(dim=1)
        f = f.sum(dim=1)
    return f, ok


class DiceMetric:
    """(pred labels, true labels) -> (mean Dice over 9 classes, per-class Dice)."""

    def __init__(self, n_classes: int = N_CLASSES):
        self.n_classes = n_classes

    def __call__(self, pred_labels: torch.Tensor, true_labels: torch.Tensor):
        p = one_hot(pred_labels.unsqueeze(1), self.n_classes)
        t = one_hot(true_labels.unsqueeze(1), self.n_classes)
        score = meandice(p, t, include_background=False)
        per_class = metric_reduction(score, "mean_batch")[0]
        return per_class.mean(), per_class


def dice_from_counts(inter, pred, true):
    """Same metric from exact integer counts (B,9) — what the GPU kernel produces."""
    inter, pred, true = (torch.as_tensor(v).to(torch.float32) for v in (inter, pred, true))
    nan = torch.tensor(float("nan"))
    score = torch.where(true > 0, 2.0 * inter / (true + pred), nan)
    per_class = metric_reduction(score, "mean_batch")[0]
    return per_class.mean(), per_class
