"""Drop-in for reference capstone/models/losses.py (+ capstone/volumetric/losses.py): the loss registry
and ``MultipleLossWrapper`` with the same names, call signature and return type (dict name -> scalar
tensor with grad), computed by the fused HIP loss pass (capstone_amd.segloss) for any spatial rank.

Deliberate fix, flagged: the reference's 3-D wrapper resolves ``"Dice"``, ``"GeneralizedDice"`` and
``"Focal"`` in the 2-D registry whose ``assert input.ndim == 4`` raises on volumes (SURVEY.md §3.1
bug 1).  Here the same formulas run on 4-D and 5-D input alike.  ``"Boundary"`` (2-D only, needs
distance maps from the CPU data pipeline, models/losses.py:124-157) is outside the hot path and raises.
"""
import torch
import torch.nn as nn

from .. import _native as nat
from .. import segloss
from ..segloss import CLASS_WEIGHT, N_CLASSES

WEIGHT = dict(zip(["Background", "BrainStem", "Chiasm", "Mandible", "OpticNerve_L", "OpticNerve_R", "Parotid_L",
                   "Parotid_R", "Submandibular_L", "Submandibular_R"], CLASS_WEIGHT))

LOSSES = {name: name for name in segloss.LOSS_NAMES}  # registry keys as in models/losses.py:160-167 (minus Boundary)


def _engine_for(logits):
    """SegLossEngine cached on the producing plan (or built ad hoc for a foreign tensor)."""
    B, C = logits.shape[:2]
    S = logits[0, 0].numel()
    plan = getattr(logits, "_ctseg_plan", None)
    holder = plan if plan is not None else logits
    eng = getattr(holder, "_ctseg_loss", None)
    if eng is None or (eng.B, eng.S, eng.C) != (B, S, C) or eng.device != logits.device:
        eng = segloss.SegLossEngine(logits.device, B, S, C)
        if plan is not None:
            plan._ctseg_loss = eng
    return eng, plan


def _as_cl(logits):
    """channels-last fp32 storage (ptr, ld, keepalive) of a (B,C,*sp) logits tensor"""
    ld = segloss.cl_logits(logits)
    if ld is not None:
        return logits.data_ptr(), ld, logits
    B, C = logits.shape[:2]
    ldn = (C + 3) // 4 * 4
    buf = torch.zeros((B, logits[0, 0].numel(), ldn), dtype=torch.float32, device=logits.device)
    buf[..., :C].copy_(logits.detach().reshape(B, C, -1).permute(0, 2, 1))
    return buf.data_ptr(), ldn, buf


class _SegLossFn(torch.autograd.Function):
    """values of the requested losses; backward writes d(sum_i g_i * loss_i)/d(logits)."""

    @staticmethod
    def forward(ctx, logits, eng, plan, names, exclude_missing, indicator):
        ptr, ld, keep = _as_cl(logits)
        eng.stats(ptr, ld, weighted_too="WeightedCrossEntropy" in names)
        vals = eng.loss_values(names, exclude_missing, indicator)
        ctx.eng, ctx.plan, ctx.names, ctx.cl = eng, plan, names, (ptr, ld, keep)
        ctx.shape = logits.shape
        return tuple(vals[n] for n in names)

    @staticmethod
    def backward(ctx, *gs):
        eng, plan = ctx.eng, ctx.plan
        ptr, ld, _ = ctx.cl
        eng.build_coef({n: (g if g is not None else 0.0) for n, g in zip(ctx.names, gs)})
        if plan is not None:   # hand the gradient to the UNet plan in its own storage dtype, no fp32 round trip
            dl = plan.dlogits
            eng.grad(ptr, ld, dl.ptr(), dl.ld, plan.dt)
            plan.dlogits_is_current = True
            g = torch.zeros((), dtype=torch.float32, device=eng.device).expand(ctx.shape)
        else:
            B, C = ctx.shape[:2]
            ldn = (C + 3) // 4 * 4
            buf = torch.zeros((B, eng.S, ldn), dtype=torch.float32, device=eng.device)
            eng.grad(ptr, ld, buf.data_ptr(), ldn, nat.F32)
            g = buf[..., :C].permute(0, 2, 1).reshape(ctx.shape)
        return g, None, None, None, None, None


class _SegLossTableFn(torch.autograd.Function):
    """the unreduced (B, C-1) / (B, C) table of ONE loss (``reduction="none"``); backward takes the table's gradient"""

    @staticmethod
    def forward(ctx, logits, eng, plan, name):
        ptr, ld, keep = _as_cl(logits)
        eng.stats(ptr, ld)
        ctx.eng, ctx.plan, ctx.name, ctx.cl, ctx.shape = eng, plan, name, (ptr, ld, keep), logits.shape
        return eng.loss_table(name)

    @staticmethod
    def backward(ctx, g):
        eng = ctx.eng
        ptr, ld, _ = ctx.cl
        eng.stats(ptr, ld)              # another entry's forward may have run on the same engine since
        eng.build_coef({ctx.name: g})
        B, C = ctx.shape[:2]
        ldn = (C + 3) // 4 * 4
        buf = torch.zeros((B, eng.S, ldn), dtype=torch.float32, device=eng.device)
        eng.grad(ptr, ld, buf.data_ptr(), ldn, nat.F32)
        return buf[..., :C].permute(0, 2, 1).reshape(ctx.shape), None, None, None


class LossEntry(nn.Module):
    """One value of ``MultipleLossWrapper.losses`` — the reference builds ``nn.ModuleDict({name: LOSSES[name](reduction=...)})``
    (capstone/models/losses.py:177-180) and code that iterates ``loss_func.losses.items()`` calls ``fx(input, target)``.
    Same call, same return: a scalar for ``reduction="mean"`` and for the two cross-entropies (whose wrappers ignore the
    argument, :45-68), the (B, C-1) Dice / GeneralizedDice or (B, C) Focal table for ``reduction="none"`` — computed by the
    fused HIP loss pass, differentiable w.r.t. ``input``."""

    def __init__(self, name, reduction="mean"):
        super().__init__()
        self.name, self.reduction = name, reduction
        if name == "WeightedCrossEntropy":
            self.weight = torch.as_tensor(list(WEIGHT.values()))     # attribute the reference's wrapper carries (:64)

    def forward(self, input, target):
        nat.require_gpu(input, f"{self.name} loss")
        eng, plan = _engine_for(input)
        stash = getattr(target, "_ctseg_labels", None)
        if stash is not None:
            eng.set_labels(*stash)
        else:
            eng.set_labels_from_i64(target)
        if self.reduction == "none" and self.name not in ("CrossEntropy", "WeightedCrossEntropy"):
            return _SegLossTableFn.apply(input, eng, None, self.name)
        return _SegLossFn.apply(input, eng, None, (self.name,), False, None)[0]

    def extra_repr(self):
        return f"{self.name}, reduction={self.reduction!r}"


class MultipleLossWrapper(nn.Module):
    def __init__(self, losses, exclude_missing=False):
        super().__init__()
        self.exclude_missing = exclude_missing
        for name in losses:
            if name == "Boundary":
                raise NotImplementedError("Boundary loss needs CPU distance maps (2-D pipeline); outside the MI355X hot path")
            assert name in LOSSES.keys()
        self.names = list(losses)
        # the reference's attribute (models/losses.py:177-180): one module per requested loss, keyed by name, built with
        # reduction "none" under exclude_missing.  forward() below computes all of them in ONE pass over the logits instead of
        # calling the entries one by one; each entry is callable on its own with the reference's (input, target) signature.
        reduction = "none" if exclude_missing else "mean"
        self.losses = nn.ModuleDict({name: LossEntry(name, reduction) for name in losses})

    def forward(self, input, target, mask_indicator=None, dist_maps=None):
        nat.require_gpu(input, "MultipleLossWrapper")
        eng, plan = _engine_for(input)
        stash = getattr(target, "_ctseg_labels", None)
        if stash is not None:
            eng.set_labels(*stash)
        else:
            eng.set_labels_from_i64(target)
        if mask_indicator is not None:
            mask_indicator = mask_indicator.type_as(input)
        vals = _SegLossFn.apply(input, eng, plan, tuple(self.names), self.exclude_missing, mask_indicator)
        return dict(zip(self.names, vals))


def apply_missing_mask(name, loss, mask_indicator):
    """models/losses.py:206-221 on a (B,C) loss table (tiny host-side table algebra on device tensors)."""
    if name == "Focal":
        bg = (mask_indicator.sum(dim=1, keepdim=True) == (N_CLASSES - 1)).float()
        mask_indicator = torch.cat([bg, mask_indicator], dim=1)
    w = 1.0 / mask_indicator.sum(dim=0)
    if torch.any(torch.isinf(w)):
        w = torch.ones_like(w)
    w = w / w.sum()
    return (loss * w[None, :] * mask_indicator).sum(dim=1).mean()
