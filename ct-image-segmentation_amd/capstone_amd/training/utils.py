"""Drop-in for the hot-path helpers of reference capstone/training/utils.py:13-20 (mixup helpers are a
2-D-only CPU-side training trick and are out of scope)."""
import torch

from .. import _native as nat
from .. import segloss


def _squash_masks(masks, n_classes, device=None):
    """2-D variant (B, K, H, W) -> (B, H, W) int64; same kernel."""
    lab_u8, lab_i64, hist = segloss.squash_masks(masks, n_classes, want_i64=True)
    lab_i64._ctseg_labels = (lab_u8, hist)
    return lab_i64


def _squash_predictions(preds):
    """softmax(dim=1).argmax(dim=1) with the reference's tie behaviour (first maximal softmax value),
    fused into one pass over the logits: (B, C, *sp) fp32 -> (B, *sp) int64."""
    nat.require_gpu(preds, "_squash_predictions")
    from ..models.losses import _as_cl
    B, C = preds.shape[:2]
    ptr, ld, keep = _as_cl(preds.float() if preds.dtype != torch.float32 else preds)
    eng = segloss.SegLossEngine(preds.device, B, preds[0, 0].numel(), C)
    out = eng.predictions(ptr, ld)
    return out.reshape((B,) + tuple(preds.shape[2:])).long()
