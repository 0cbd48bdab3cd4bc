"""Loss / metric engine: host side of ctseg_squash_masks / ctseg_seg_loss / ctseg_dice_counts.

One pass over the fp32 channels-last logits produces every sum the reference's losses and its Dice
metric need (capstone/models/losses.py, capstone/models/temp.py, capstone/models/metrics.py); the
per-(sample, class) closed forms are then a handful of ops on (B, 10) device tensors, and one more
pass writes d(loss)/d(logits).  CE-only (the reference's 3-D default, base_trainer.py:28) is a single
fused pass.
"""
import math

import torch

from . import _native as nat
from ._native import BF16, F32

N_CLASSES = 10
CLASS_WEIGHT = (1e-10, 0.007, 0.3296, 0.0046, 0.2619, 0.3035, 0.0068, 0.0065, 0.0374, 0.0426)  # models/losses.py:10-21
SMOOTH = 1e-5
LOSS_NAMES = ("CrossEntropy", "WeightedCrossEntropy", "Focal", "Dice", "GeneralizedDice")


def cl_logits(t):
    """(B,C,*sp) fp32 tensor -> (channels-last storage tensor [B,S,ld] sharing memory, ld) or None"""
    if t.dtype != torch.float32 or not t.is_cuda or t.ndim < 3:
        return None
    perm = (0,) + tuple(range(2, t.ndim)) + (1,)
    v = t.permute(perm)
    st, sh = v.stride(), v.shape
    if st[-1] != 1:
        return None
    ld = st[-2] if v.ndim > 2 else sh[-1]
    exp = ld
    for d in range(v.ndim - 2, -1, -1):   # every outer dim must be dense over ld-strided voxels
        if st[d] != exp and sh[d] != 1:
            return None
        exp *= sh[d]
    if ld % 4 or ld < sh[-1] or ld > 16 or t.data_ptr() % 16:
        return None
    return ld


def squash_masks(masks, n_classes=N_CLASSES, want_i64=True):
    """_squash_masks_3D / _squash_masks on device: (B,K,*sp) uint8 -> labels u8 (B,S), int64 (B,*sp), hist (B,K+1)."""
    nat.require_gpu(masks, "_squash_masks")
    pre = getattr(masks, "_ctseg_labels", None)
    if pre is not None and masks.dtype == torch.uint8:
        # label maps squashed by the device input pipeline (volumetric/datasets.collate_3d): nothing left to do
        return pre[0], (masks.long() if want_i64 else None), pre[1]
    if masks.dtype != torch.uint8:
        masks = masks.to(torch.uint8)
    masks = masks.contiguous()
    B, K = masks.shape[:2]
    S = masks[0, 0].numel()
    assert K == n_classes - 1
    lab = torch.empty((B, S), dtype=torch.uint8, device=masks.device)
    lab64 = torch.empty((B,) + tuple(masks.shape[2:]), dtype=torch.int64, device=masks.device) if want_i64 else None
    hist = torch.zeros((B, K + 1), dtype=torch.int64, device=masks.device)
    nat.call("ctseg_squash_masks", masks.data_ptr(), B, K, S, lab.data_ptr(), nat.ptr(lab64), hist.data_ptr())
    return lab, lab64, hist


class SegLossEngine:
    def __init__(self, device, B, S, C=N_CLASSES):
        self.device, self.B, self.S, self.C = device, B, S, C
        self.P = max(1, min(2048, math.ceil(S / 2048)))
        self.R = 2 + 3 * C
        f64 = dict(dtype=torch.float64, device=device)
        self.part = torch.zeros((B, self.P, self.R), **f64)
        self.red = torch.zeros((B, self.R), **f64)
        self.red_w = torch.zeros((B, self.R), **f64)
        self.cnt = torch.zeros((B, 3, C), dtype=torch.int64, device=device)
        self.coef = torch.zeros((B, 1 + 3 * C), dtype=torch.float32, device=device)
        self.cw = torch.tensor(CLASS_WEIGHT[:C], dtype=torch.float32, device=device)
        self.cw_eff = torch.ones(C, dtype=torch.float32, device=device)
        self.labels = self.hist = None
        self.own_dlogits = {}

    # ---- targets ----
    def set_labels(self, labels_u8, hist):
        self.labels, self.hist = labels_u8, hist

    def set_labels_from_i64(self, target):
        t = target.reshape(self.B, self.S)
        self.labels = t.to(torch.uint8).contiguous()
        self.hist = torch.zeros((self.B, self.C), dtype=torch.int64, device=self.device)
        self.hist.scatter_add_(1, t.long(), torch.ones_like(t, dtype=torch.int64))

    # ---- passes ----
    def _pass(self, logits_ptr, ld, cw, do_stats, do_grad, coef=None, dl_ptr=None, g_ld=0, gdt=F32, pred=None, part=None):
        """do_stats / do_grad: False, True, or 2 = cross-entropy-only fast path of the kernel"""
        nat.call("ctseg_seg_loss", logits_ptr, ld, self.labels.data_ptr(), self.B, self.S, self.C, nat.ptr(cw),
                 int(do_stats), nat.ptr(part if part is not None else self.part), self.P, self.cnt.data_ptr(),
                 int(do_grad), nat.ptr(coef), dl_ptr, g_ld, gdt, nat.ptr(pred))

    def stats(self, logits_ptr, ld, weighted_too=False):
        self.cnt.zero_()
        self._pass(logits_ptr, ld, None, True, False)
        nat.call("ctseg_reduce_partials_f64", self.part.data_ptr(), self.B, self.P, self.R, self.red.data_ptr())
        if weighted_too:
            keep = self.cnt.clone()
            self._pass(logits_ptr, ld, self.cw, True, False)
            nat.call("ctseg_reduce_partials_f64", self.part.data_ptr(), self.B, self.P, self.R, self.red_w.data_ptr())
            self.cnt.copy_(keep)

    def prepare_fused_ce(self, weighted=False):
        """the part of fused_ce that needs only the label histogram (counters zeroed, 1/denominator table): the training step
        issues it on the side stream right behind the mask squash, off the forward -> loss critical path"""
        self.cnt.zero_()
        w = self.cw.double() if weighted else torch.ones(self.C, dtype=torch.float64, device=self.device)
        denom = (self.hist.double() * w[None, :]).sum()
        self.coef.zero_()
        self.coef[:, 0] = (1.0 / denom).float()
        self._ce_ready = weighted

    def fused_ce(self, logits_ptr, ld, dl_ptr, g_ld, gdt, weighted=False):
        """CrossEntropy (or WeightedCrossEntropy) alone: stats + gradient in ONE pass (upstream gradient 1)."""
        if getattr(self, "_ce_ready", None) is not weighted:
            self.prepare_fused_ce(weighted)
        self._ce_ready = None
        self._pass(logits_ptr, ld, self.cw if weighted else None, 2, 2, self.coef, dl_ptr, g_ld, gdt)
        nat.call("ctseg_reduce_partials_f64", self.part.data_ptr(), self.B, self.P, self.R,
                 (self.red_w if weighted else self.red).data_ptr())

    def head_ce(self, desc, slots, dl_ptr, g_ld, weighted=False):
        """the logits convolution and the cross-entropy (loss sums, Dice counts, d loss / d logits) in ONE launch: ``desc`` is the
        recorded descriptor of the plan's last forward op, ``slots`` = plan.head_ce_slots().  Same tables as fused_ce."""
        if getattr(self, "_ce_ready", None) is not weighted:
            self.prepare_fused_ce(weighted)
        self._ce_ready = None
        part = getattr(self, "_part_head", None)
        if part is None or part.shape[1] != slots:
            part = self._part_head = torch.zeros((self.B, slots, self.R), dtype=torch.float64, device=self.device)
        nat.call("ctseg_conv_logits_ce", desc, self.labels.data_ptr(), self.C, nat.ptr(self.cw if weighted else None),
                 self.coef.data_ptr(), self.coef.shape[1], dl_ptr, g_ld, part.data_ptr(), slots, self.R, self.cnt.data_ptr())
        nat.call("ctseg_reduce_partials_f64", part.data_ptr(), self.B, slots, self.R, (self.red_w if weighted else self.red).data_ptr())

    def ce_summary(self, weighted=False):
        """(loss, mean Dice, Dice per class) of the last fused_ce pass as views of one small tensor: one launch instead of the
        ~20 scalar-sized torch kernels of loss_values() + dice_metric()"""
        out = torch.empty(1 + self.C, dtype=torch.float32, device=self.device)
        nat.call("ctseg_loss_dice_summary", (self.red_w if weighted else self.red).data_ptr(), self.B, self.R, self.cnt.data_ptr(),
                 self.C, out.data_ptr())
        self.last_summary = out
        return out[0], out[1], out[2:]

    def predictions(self, logits_ptr, ld):
        pred = torch.empty((self.B, self.S), dtype=torch.uint8, device=self.device)
        if self.labels is None:
            self.labels = torch.zeros((self.B, self.S), dtype=torch.uint8, device=self.device)
        self._pass(logits_ptr, ld, None, False, False, pred=pred)
        return pred

    # ---- closed forms on (B, C) tables ----
    def _tables(self):
        C = self.C
        r = self.red
        return r[:, 2:2 + C], r[:, 2 + C:2 + 2 * C], r[:, 2 + 2 * C:2 + 3 * C], self.hist.double()

    def loss_values(self, names, exclude_missing=False, indicator=None):
        """dict name -> 0-dim fp32 tensor, plus per-loss (B,C) weights used for the gradient tables."""
        B, C, S = self.B, self.C, self.S
        P, I, FO, Y = self._tables()
        out, self._w = {}, {}
        for name in names:
            if name == "CrossEntropy":
                out[name] = (self.red[:, 0].sum() / self.red[:, 1].sum()).float()
            elif name == "WeightedCrossEntropy":
                out[name] = (self.red_w[:, 0].sum() / self.red_w[:, 1].sum()).float()
            elif name in ("Dice", "GeneralizedDice"):
                if name == "Dice":
                    f = 1.0 - (2.0 * I + SMOOTH) / (P + Y + SMOOTH)
                else:
                    w = self._gdl_w(Y)
                    f = 1.0 - (2.0 * I * w + SMOOTH) / ((P + Y) * w + SMOOTH)
                f = f[:, 1:].float()
                wt = self._mask_weights(name, indicator, exclude_missing, C - 1)       # (B, C-1)
                out[name] = (f * wt).sum()
                self._w[name] = torch.cat([torch.zeros_like(wt[:, :1]), wt], 1)
            elif name == "Focal":
                f = (FO / S).float()
                wt = self._mask_weights(name, indicator, exclude_missing, C)
                out[name] = (f * wt).sum()
                self._w[name] = wt
            else:
                raise KeyError(name)
        return out

    def loss_table(self, name):
        """the unreduced per-(sample, class) table of one loss after ``stats()`` — what the reference's wrapper returns with
        ``reduction="none"`` (capstone/models/losses.py:177-180 builds every entry that way under ``exclude_missing``):
        Dice / GeneralizedDice (B, C-1) (background excluded), Focal (B, C)"""
        P, I, FO, Y = self._tables()
        if name == "Dice":
            return (1.0 - (2.0 * I + SMOOTH) / (P + Y + SMOOTH))[:, 1:].float()
        if name == "GeneralizedDice":
            w = self._gdl_w(Y)
            return (1.0 - (2.0 * I * w + SMOOTH) / ((P + Y) * w + SMOOTH))[:, 1:].float()
        if name == "Focal":
            return (FO / self.S).float()
        raise KeyError(name)

    def _table_weight(self, name, g):
        """weight of table entry (b, c) in the scalar being differentiated: ``_w[name] * g`` for the reduced losses, or the
        upstream gradient itself when the caller took the unreduced table (``g`` is then (B, C-1) or (B, C))"""
        g = torch.as_tensor(g, dtype=torch.float64, device=self.device)
        if g.ndim == 2:
            if g.shape[1] == self.C - 1:
                g = torch.cat([torch.zeros_like(g[:, :1]), g], 1)
            return g
        return self._w[name].double() * g

    def _gdl_w(self, Y):
        y = Y.float()
        w = torch.reciprocal(y * y)                      # models/temp.py:90-94 (Weight.SQUARE)
        w = w[:, 1:]
        inf = torch.isinf(w)
        w = torch.where(inf, torch.zeros_like(w), w)
        w = torch.where(inf, w.max(dim=1, keepdim=True).values.expand_as(w), w)  # temp.py:150-153
        return torch.cat([torch.ones_like(y[:, :1]), w], 1).double()

    def _mask_weights(self, name, indicator, exclude_missing, ncol):
        """weight of table entry (b,c) in the scalar loss: plain mean, or models/losses.py:206-221"""
        B = self.B
        if not exclude_missing:
            return torch.full((B, ncol), 1.0 / (B * ncol), dtype=torch.float32, device=self.device)
        ind = indicator.float()
        if name == "Focal":
            bg = (ind.sum(dim=1, keepdim=True) == (self.C - 1)).float()
            ind = torch.cat([bg, ind], dim=1)
        w = 1.0 / ind.sum(dim=0)
        w = torch.where(torch.isinf(w).any(), torch.ones_like(w), w)
        w = w / w.sum()
        return w[None, :] * ind / B

    def build_coef(self, scales):
        """scales: dict name -> upstream gradient (0-dim tensor or float). Fills coef (B,1+3C) and cw_eff (C)."""
        B, C, S = self.B, self.C, self.S
        P, I, FO, Y = self._tables()
        dev = self.device
        a = torch.zeros((B, C), dtype=torch.float64, device=dev)
        b = torch.zeros((B, C), dtype=torch.float64, device=dev)
        fcoef = torch.zeros((B, C), dtype=torch.float64, device=dev)
        cw = torch.zeros(C, dtype=torch.float64, device=dev)
        for name, g in scales.items():
            g = torch.as_tensor(g, dtype=torch.float64, device=dev)     # 0-dim, or the (B, C[-1]) gradient of an unreduced table
            if name == "CrossEntropy":
                cw = cw + g / float(B * S)
            elif name == "WeightedCrossEntropy":
                cw = cw + g * self.cw.double() / (self.hist.double() * self.cw.double()[None, :]).sum()
            elif name == "Dice":
                D = P + Y + SMOOTH
                wt = self._table_weight(name, g)
                a = a - 2.0 * wt / D
                b = b + wt * (2.0 * I + SMOOTH) / (D * D)
            elif name == "GeneralizedDice":
                w = self._gdl_w(Y)
                D = (P + Y) * w + SMOOTH
                wt = self._table_weight(name, g)
                a = a - 2.0 * w * wt / D
                b = b + wt * w * (2.0 * I * w + SMOOTH) / (D * D)
            elif name == "Focal":
                fcoef = fcoef + self._table_weight(name, g) / float(S)
        self.cw_eff.copy_(cw.float())
        self.coef[:, 0] = 1.0
        self.coef[:, 1:1 + C] = a.float()
        self.coef[:, 1 + C:1 + 2 * C] = b.float()
        self.coef[:, 1 + 2 * C:] = fcoef.float()

    def grad(self, logits_ptr, ld, dl_ptr, g_ld, gdt):
        self._pass(logits_ptr, ld, self.cw_eff, False, True, self.coef, dl_ptr, g_ld, gdt)

    # ---- Dice metric from exact integer counts (models/metrics.py:15-21, temp.py:173-292) ----
    def dice_metric(self, cnt=None):
        cnt = self.cnt if cnt is None else cnt
        inter, pred, true = (cnt[:, i, 1:].float() for i in range(3))
        nan = torch.full_like(inter, float("nan"))
        score = torch.where(true > 0, 2.0 * inter / (true + pred), nan)      # (B, 9), NaN where truth empty
        ok = (~torch.isnan(score)).float()
        score = torch.nan_to_num(score, nan=0.0)
        n_ok = ok.sum(dim=0)
        per_class = torch.where(n_ok > 0, score.sum(dim=0) / n_ok, torch.zeros_like(n_ok))   # "mean_batch"
        return per_class.mean(), per_class


def dice_counts(pred_u8, true_u8, C=N_CLASSES):
    B, S = pred_u8.shape
    cnt = torch.zeros((B, 3, C), dtype=torch.int64, device=pred_u8.device)
    nat.call("ctseg_dice_counts", pred_u8.data_ptr(), true_u8.data_ptr(), B, S, C, cnt.data_ptr())
    return cnt
