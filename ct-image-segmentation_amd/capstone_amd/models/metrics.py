"""Drop-in for reference capstone/models/metrics.py (+ capstone/volumetric/metrics.py:5-20).

``DiceMetricWrapper()(pred_labels, true_labels) -> (mean Dice, per-class Dice (9,))`` with the reference's
aggregation (per-sample per-class Dice, NaN where the class is absent from the truth; NaN-aware batch
mean per class; mean over the 9 classes — models/temp.py:173-214, :271-273; models/metrics.py:15-21).
The reference builds two (B,10,*sp) fp32 one-hots and their product; here exact integer counts come
from one pass over the two label maps (ctseg_dice_counts) — or for free from the fused loss pass.
"""
import torch

from .. import _native as nat
from .. import segloss
from ..segloss import N_CLASSES


class DiceMetricWrapper(object):
    def __init__(self):
        self.n_classes = N_CLASSES

    def __call__(self, input, target):
        nat.require_gpu(input, "DiceMetricWrapper")
        B = input.shape[0]
        cnt = getattr(input, "_ctseg_counts", None)   # predictions that came out of the fused loss pass
        if cnt is None:
            p = input.reshape(B, -1).to(torch.uint8).contiguous()
            t = target.reshape(B, -1).to(torch.uint8).contiguous()
            cnt = segloss.dice_counts(p, t, self.n_classes)
        self.last_counts = cnt      # (B, 3, C) int64: what epoch_dice_across_ranks gathers at N > 1
        eng = segloss.SegLossEngine.__new__(segloss.SegLossEngine)
        return segloss.SegLossEngine.dice_metric(eng, cnt)
