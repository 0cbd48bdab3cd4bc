"""Drop-in for reference capstone/volumetric/base_trainer.py: ``BaseUNet3D`` with the same constructor,
``forward`` / ``training_step`` / ``validation_step`` / ``_shared_step`` / ``configure_optimizers`` /
``add_model_specific_args`` surface and hyper-parameter names, running on the MI355X engine.

Works as a ``pytorch_lightning.LightningModule`` when Lightning is importable and as a plain
``nn.Module`` otherwise (Lightning is not in this image); ``fit_step`` is the native step that
reproduces Lightning 1.0's per-batch order (zero_grad -> training_step -> backward -> [DDP mean of
gradients] -> Adam.step) without autograd, and is what bench.py times.
"""
from argparse import ArgumentParser
from typing import List

import torch
import torch.nn as nn

from .. import STRUCTURES, segloss
from .. import _native as nat
from ..models import UNet
from ..training.utils import _squash_predictions  # noqa: F401  (same import the reference has)
from .losses import MultipleLossWrapper3D
from .metrics import DiceMetricWrapper3D
from .utils import _squash_masks_3D

try:  # pragma: no cover - Lightning is absent from the build image
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:  # noqa: BLE001
    pl = None

    class _HParams(dict):
        __getattr__ = dict.__getitem__

    class _Base(nn.Module):
        """The slice of LightningModule the reference's module uses."""

        def __init__(self):
            super().__init__()
            self.hparams = _HParams()
            self.logged = {}
            self._epoch = {}

        def save_hyperparameters(self, *names, frame_locals=None):
            for n in names:
                self.hparams[n] = frame_locals.get(n, frame_locals.get("kwargs", {}).get(n))

        def log(self, name, value, on_step=False, on_epoch=True, **kw):
            """``logged[name]`` = the value of the last step; with ``on_epoch=True`` (what the reference passes everywhere,
            volumetric/base_trainer.py:106-109,125-131) the value also enters a running sum whose mean over the steps of the
            epoch is what Lightning 1.0 reports (``epoch_means``)."""
            v = value.detach() if torch.is_tensor(value) else value
            self.logged[name] = v
            if on_epoch:
                acc = self._epoch.get(name)
                if acc is None:
                    self._epoch[name] = [v.clone() if torch.is_tensor(v) else v, 1]
                else:
                    acc[0] = acc[0] + v
                    acc[1] += 1

        def log_packed(self, packed, layout):
            """several logged values that are slices of ONE small device tensor (the native step's loss / Dice summary):
            one accumulate for all of them instead of one tiny launch per name.  layout: [(name, index or slice)]"""
            packed = packed.detach()
            for name, sl in layout:
                self.logged[name] = packed[sl]
            key = ("packed",) + tuple(n for n, _ in layout)
            acc = self._epoch.get(key)
            if acc is None:
                self._epoch[key] = [packed.clone(), 1, layout]
            else:
                acc[0] += packed
                acc[1] += 1

        def epoch_means(self, reset=True):
            """{name: mean over the steps logged since the last reset} — Lightning's ``on_epoch=True`` reduction"""
            out = {}
            for k, acc in self._epoch.items():
                if isinstance(k, tuple):
                    mean = acc[0] / acc[1]
                    out.update({name: mean[sl] for name, sl in acc[2]})
                else:
                    out[k] = acc[0] / acc[1]
            if reset:
                self._epoch = {}
            return out

        @property
        def device(self):
            return next(self.parameters()).device

SEED = 12342


class _DataParallelSurface:
    """What both drop-in modules (BaseUNet3D, BaseUNet2D) need so that the reference's ONLY way into multi-GPU training —
    ``Trainer.from_argparse_args(args)`` with ``--gpus N --distributed_backend ddp`` (capstone/volumetric/base_trainer.py:196,217;
    capstone/training/base_trainer.py: the same Trainer flags) — lands on the engine's own gradient exchange."""

    # ---- the gradient reducer lives on the engine (plan.Engine.reducer): one object for every route into the backward ----
    @property
    def reducer(self):
        return self.unet.engine().reducer

    @reducer.setter
    def reducer(self, value):
        self.unet.engine().reducer = value

    def configure_ddp(self, model, device_ids):
        """Lightning 1.0 hook (``LightningModule.configure_ddp(model, device_ids)``, called by the DDP accelerator once the process
        group exists and the module sits on its GPU; stock: ``LightningDistributedDataParallel(model, device_ids, ...)``).  Here:
        build the flat parameter store on the module's device, make the replicas identical and install the flat-buffer gradient
        exchange (``distributed.attach``), and return a pass-through wrapper with DDP's ``.module`` / per-mode ``forward``."""
        from .. import distributed as cdist
        dev = next(model.unet.parameters()).device
        model.unet.engine().ensure(dev)
        cdist.attach(model)
        return cdist.NativeDataParallel(model, device_ids)

    @property
    def _ddp_params_and_buffers_to_ignore(self):
        """torch's ``DistributedDataParallel.__init__`` probes the wrapped module for this attribute: the probe is the tripwire.
        A stock DDP wrap would train on rank-local gradients without any error (its reducer waits for AccumulateGrad hooks that
        never fire: the HIP kernels write the flat gradient buffer and the autograd nodes return None for the parameters)."""
        raise nat.NativeError(
            "torch.nn.parallel.DistributedDataParallel cannot wrap this module: its gradients are written into one flat buffer by "
            "the HIP backward kernels and averaged by capstone_amd.distributed.attach(module). Under Lightning the module's own "
            "configure_ddp() does that; in a hand-written loop call attach(module) after init_process_group and skip the DDP wrap.")

    # ---- Dice across ranks (the integer counts, one small all-gather at epoch end) -------------------------
    def _keep_dice_counts(self, cnt, prefix):
        """remember this step's per-sample integer Dice counts (B, 3, C) when the module is data-parallel (or on request,
        ``CTSEG_KEEP_DICE_COUNTS=1``): 30 int64 per sample and step"""
        import os
        red = self.unet.engine().reducer
        if (red is not None and red.world > 1) or os.environ.get("CTSEG_KEEP_DICE_COUNTS") == "1":
            self.__dict__.setdefault("_dice_counts", {}).setdefault(prefix, []).append(cnt.detach().clone())

    def epoch_dice_across_ranks(self, prefix="train", reset=True, group=None):
        """(mean Dice, per-class Dice (9,)) of the epoch as the reference computes it on the GLOBAL batch: per step, the
        per-sample counts of all ranks are one batch for ``compute_meandice`` + ``do_metric_reduction("mean_batch")``
        (capstone/models/temp.py:173-292, capstone/models/metrics.py:15-21); the epoch value is the mean over the steps
        (Lightning's ``on_epoch=True``).  The reference itself logs per rank (no ``sync_dist``, :106-109,125-131); this is the
        figure "Dice vs ref" is read from at N > 1.  One all-gather of (steps, B, 3, C) int64.  None when nothing was kept."""
        from .. import distributed as cdist
        kept = self.__dict__.get("_dice_counts", {}).get(prefix)
        if not kept:
            return None
        if len({tuple(k.shape) for k in kept}) == 1:
            cnt = list(cdist.gather_dice_counts(torch.stack(kept), group))  # ONE collective: (steps, world * B, 3, C)
        else:                                                               # a short last batch: one small collective per step
            cnt = [cdist.gather_dice_counts(k, group) for k in kept]
        if reset:
            self._dice_counts[prefix] = []
        eng = segloss.SegLossEngine.__new__(segloss.SegLossEngine)
        per_step = [segloss.SegLossEngine.dice_metric(eng, c) for c in cnt]
        mean = torch.stack([m for m, _ in per_step]).mean()
        per_class = torch.stack([p for _, p in per_step]).mean(dim=0)
        return mean, per_class


def _precision(kwargs):
    """Lightning's Trainer flag arrives through **vars(args) too.  ``--precision 16`` is IEEE half, as in the reference's stack
    (Lightning 1.0 native AMP): fp16 STORAGE for the inference passes (validation / test / sliding window), and bf16 storage
    for the training plans after a one-time RuntimeWarning (plan.Engine.train_dt: no fp16 backward kernels, no loss scaling;
    bf16 has fp32's range).  "bf16" by name uses bf16 throughout.  Nothing is remapped silently."""
    p = kwargs.get("precision", "fp32")
    if p in (16, "16", "fp16", "16-mixed", "16-true"):
        return "fp16"
    if p in ("bf16", "bf16-mixed", "bf16-true"):
        return "bf16"
    if p in (32, "32", "fp32", "32-true", None):
        return "fp32"
    raise ValueError(f"unsupported precision {p!r}")


class BaseUNet3D(_DataParallelSurface, _Base):
    def __init__(self, filters: List = [16, 32, 64, 128, 256], use_res_units: bool = False, downsample: bool = False,
                 lr: float = 1e-3, loss_fx: list = ["CrossEntropy"], exclude_missing: bool = False, **kwargs) -> None:
        super().__init__()
        assert isinstance(loss_fx, list), "This module expects a list of loss functions"
        loss_fx.sort()  # consistent order of loss functions (reference :35)
        names = ("batch_size", "transform_degree", "filters", "use_res_units", "downsample", "lr", "loss_fx", "exclude_missing")
        if pl is not None:
            self.save_hyperparameters(*names)
        else:
            self.save_hyperparameters(*names, frame_locals=dict(locals()))
        self._precision = _precision(kwargs)
        self.unet = self._construct_model()
        self.loss_func = MultipleLossWrapper3D(losses=loss_fx, exclude_missing=exclude_missing)
        self.dice_score = DiceMetricWrapper3D()

    @property
    def _n_classes(self):
        return len(STRUCTURES) + 1

    def _construct_model(self):
        # reference :58-72 — in_channels=1, strides for a 5-entry filter list, num_res_units hard-wired to 2
        return UNet(dimensions=3, in_channels=1, out_channels=self._n_classes, channels=self.hparams.filters,
                    strides=[2, 2, 2, 2], num_res_units=2, precision=self._precision)

    def forward(self, x):
        return self.unet(x)

    def training_step(self, batch, batch_idx=0):
        """reference :80-82.  Only the loss leaves this function, so when the step is cross-entropy-only (the reference's working
        3-D default, :28) the loss runs fused with the logits convolution (bf16: ``ctseg_conv_logits_ce``; otherwise the one-pass
        cross-entropy over the materialised logits) and the returned scalar carries an autograd node (plan._StepLossFn) whose
        backward is the recorded backward program: ``training_step -> loss.backward() -> optimizer.step()`` then does the GPU
        work of ``fit_step``.  ``_shared_step`` (which must return the prediction) keeps the two-pass route;
        ``CTSEG_DROPIN_FUSED=0`` forces it here too."""
        if self._fused_training_ok(batch):
            return self._fused_training_step(batch)
        _, _, _, _, loss = self._shared_step(batch, is_training=True)
        return loss

    def _ce_only(self):
        names = list(self.loss_func.names)
        return len(names) == 1 and names[0] in ("CrossEntropy", "WeightedCrossEntropy")

    def _fused_training_ok(self, batch):
        import os
        return (torch.is_grad_enabled() and self._ce_only() and os.environ.get("CTSEG_DROPIN_FUSED", "1") != "0"
                and any(p.requires_grad for p in self.unet.parameters()))

    def _forward_and_ce(self, batch, keep_logits):
        """squash masks (side stream) -> forward -> cross-entropy + Dice counts + d loss / d logits for an upstream gradient of 1.
        Returns (engine, plan, loss engine, weighted)."""
        images, masks, mask_indicator = batch
        eng = self.unet.engine()
        plan = eng.plan_for(images)
        side = plan.side_stream()
        le = getattr(plan, "_ctseg_loss", None)
        if le is None:
            le = plan._ctseg_loss = segloss.SegLossEngine(images.device, images.shape[0], plan.logits.S, self._n_classes)
        weighted = self.loss_func.names[0] != "CrossEntropy"
        head_slots = 0 if keep_logits else plan.head_ce_slots(self._n_classes)
        if side is not None:
            # the label map is not needed before the loss: squash the masks (and build the loss tables that need only the label
            # histogram) on the side stream while the forward pass starts
            main = torch.cuda.current_stream(images.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                lab_u8, _, hist = segloss.squash_masks(masks, self._n_classes, want_i64=False)
                le.set_labels(lab_u8, hist)
                le.prepare_fused_ce(weighted=weighted)
            for t in (lab_u8, hist):
                t.record_stream(main)
            logits = plan.forward(images, skip_head=head_slots > 0)
            main.wait_stream(side)
        else:
            lab_u8, _, hist = segloss.squash_masks(masks, self._n_classes, want_i64=False)
            le.set_labels(lab_u8, hist)
            logits = plan.forward(images, skip_head=head_slots > 0)
        dl = plan.dlogits
        if head_slots > 0:
            le.head_ce(plan._head_ce[2], head_slots, dl.ptr(), dl.ld, weighted=weighted)
        else:
            le.fused_ce(logits.ptr(), logits.ld, dl.ptr(), dl.ld, plan.dt, weighted=weighted)
        return eng, plan, le, weighted

    def _log_ce_summary(self, le, weighted, prefix="train"):
        """loss, mean Dice and the nine per-structure Dice scores of the fused pass: ONE launch, logged under the reference's names"""
        name = self.loss_func.names[0]
        loss, dm, dpc = le.ce_summary(weighted=weighted)
        if pl is None:
            self.log_packed(le.last_summary, [(f"{name} Loss ({prefix})", 0), (f"Mean Dice Score ({prefix})", 1),
                                              (f"Dice per class ({prefix})", slice(2, None))] +
                            [(f"{s_} Dice ({prefix})", 2 + i) for i, s_ in enumerate(STRUCTURES)])
        else:   # Lightning logs scalars: one entry per structure, as the reference's _log_dice_scores does (:125-129)
            self.log(f"{name} Loss ({prefix})", loss, on_step=False, on_epoch=True)
            for structure, score in zip(STRUCTURES, dpc):
                self.log(f"{structure} Dice ({prefix})", score, on_step=False, on_epoch=True)
            self.log(f"Mean Dice Score ({prefix})", dm, on_step=False, on_epoch=True)
        return loss

    def _fused_training_step(self, batch):
        from ..plan import _StepLossFn
        import os
        nat.require_gpu(batch[0], "training_step")
        eng, plan, le, weighted = self._forward_and_ce(batch, keep_logits=os.environ.get("CTSEG_DROPIN_KEEP_LOGITS", "0") == "1")
        plan.dlogits_is_current = True
        loss = self._log_ce_summary(le, weighted)
        self._keep_dice_counts(le.cnt, "train")
        return _StepLossFn.apply(loss, eng, plan, self, *eng.store.params)

    def validation_step(self, batch, batch_idx=0):
        self._shared_step(batch, is_training=False)

    def _shared_step(self, batch, is_training: bool):
        (images, masks, mask_indicator) = batch
        masks = _squash_masks_3D(masks, self._n_classes, self.device)
        mask_indicator = mask_indicator.type_as(images)
        prefix = "train" if is_training else "val"
        prediction = self.forward(images)
        prediction._ctseg_plan = self.unet.engine().last_plan
        loss_dict = self.loss_func(input=prediction, target=masks, mask_indicator=mask_indicator)
        total_loss = torch.stack(list(loss_dict.values())).sum()
        for name, loss_value in loss_dict.items():
            self.log(f"{name} Loss ({prefix})", loss_value, on_step=False, on_epoch=True)
        self._log_dice_scores(prediction, masks, mask_indicator, prefix)
        return images, masks, mask_indicator, prediction, total_loss

    def configure_optimizers(self):
        # reference :113-114 ``optim.Adam(self.parameters(), lr=self.hparams.lr)``: the same optimizer class and arguments; its
        # step() is one ctseg_adam_step launch over the flat parameter buffer (capstone_amd.optim.Adam IS a torch.optim.Adam)
        from ..optim import Adam
        return Adam(self.parameters(), lr=self.hparams.lr, unet=self.unet)

    def _log_dice_scores(self, prediction, masks, mask_indicator, prefix):
        # reference :116-132 clones 1 GB of logits, softmaxes, argmaxes and one-hots twice; the fused loss pass
        # already counted |pred==c & true==c|, |pred==c|, |true==c| with the same softmax->argmax tie rule.
        with torch.no_grad():
            eng = getattr(prediction._ctseg_plan, "_ctseg_loss", None)
            dice_mean, dice_per_class = eng.dice_metric()
            self._keep_dice_counts(eng.cnt, prefix)
            for structure, score in zip(STRUCTURES, dice_per_class):
                self.log(f"{structure} Dice ({prefix})", score, on_step=False, on_epoch=True)
            self.log(f"Mean Dice Score ({prefix})", dice_mean, on_step=False, on_epoch=True)

    # ---- native step: what Lightning's loop does per batch, without autograd -------------------------
    def fit_step(self, batch, betas=(0.9, 0.999), eps=1e-8, keep_logits=True):
        """keep_logits=False lets a cross-entropy-only step run the logits convolution with the loss fused into its epilogue
        (ctseg_conv_logits_ce): the fp32 logits are then never materialised and ``engine.logits_view()`` raises after the step;
        loss, Dice metric, gradients and the update are the same (gradient of the loss w.r.t. the logits bit-identical)."""
        images, masks, mask_indicator = batch
        nat.require_gpu(images, "fit_step")
        names = list(self.loss_func.names)
        ce_only = self._ce_only()
        vals = None
        if ce_only:
            eng, plan, le, weighted = self._forward_and_ce(batch, keep_logits)
        else:
            eng = self.unet.engine()
            plan = eng.plan_for(images)
            side = plan.side_stream()
            le = getattr(plan, "_ctseg_loss", None)
            if le is None:
                le = plan._ctseg_loss = segloss.SegLossEngine(images.device, images.shape[0], plan.logits.S, self._n_classes)
            if side is not None:
                main = torch.cuda.current_stream(images.device)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    lab_u8, _, hist = segloss.squash_masks(masks, self._n_classes, want_i64=False)
                    le.set_labels(lab_u8, hist)
                for t in (lab_u8, hist):
                    t.record_stream(main)
                logits = plan.forward(images)
                main.wait_stream(side)
            else:
                lab_u8, _, hist = segloss.squash_masks(masks, self._n_classes, want_i64=False)
                le.set_labels(lab_u8, hist)
                logits = plan.forward(images)
            dl = plan.dlogits
            le.stats(logits.ptr(), logits.ld, weighted_too="WeightedCrossEntropy" in names)
            vals = le.loss_values(names, self.loss_func.exclude_missing, mask_indicator.float())
            le.build_coef({n: 1.0 for n in names})
            le.grad(logits.ptr(), logits.ld, dl.ptr(), dl.ld, plan.dt)
        book = {}

        def bookkeeping():     # scalar-sized work: queued behind the backward pass, beside the side stream's tail
            if ce_only:        # one launch for loss + Dice
                loss, dm, dpc = le.ce_summary(weighted=names[0] != "CrossEntropy")
                book["vals"], book["total"], book["dice"], book["packed"] = {names[0]: loss}, loss, (dm, dpc), le.last_summary
                return
            book["vals"] = vals
            book["total"] = torch.stack([vals[n] for n in names]).sum()
            book["dice"] = le.dice_metric()

        self._keep_dice_counts(le.cnt, "train")        # (queued before the next step overwrites the counters)

        reducer = eng.reducer
        if reducer is not None:
            plan.backward(reducer.hooks(plan), before_join=bookkeeping)
            scale = reducer.finish()
        else:
            eng.warn_if_unattached()
            plan.backward(before_join=bookkeeping)
            scale = 1.0
        eng.store.adam_step(self.hparams.lr, betas, eps, grad_scale=scale)
        plan.repack_after_update()
        vals, total = book["vals"], book["total"]
        dice_mean, dice_per_class = book["dice"]
        if "packed" in book and pl is None:
            self.log_packed(book["packed"], [(f"{names[0]} Loss (train)", 0), ("Mean Dice Score (train)", 1),
                                             ("Dice per class (train)", slice(2, None))] +
                            [(f"{s_} Dice (train)", 2 + i) for i, s_ in enumerate(STRUCTURES)])
        else:
            for n in names:
                self.log(f"{n} Loss (train)", vals[n], on_step=False, on_epoch=True)
            self.log("Mean Dice Score (train)", dice_mean, on_step=False, on_epoch=True)
            if pl is None:
                self.log("Dice per class (train)", dice_per_class, on_step=False, on_epoch=True)
            else:   # Lightning logs scalars: one entry per structure, as the reference's _log_dice_scores does (:125-129)
                for structure, score in zip(STRUCTURES, dice_per_class):
                    self.log(f"{structure} Dice (train)", score, on_step=False, on_epoch=True)
        return total

    # ---- optimizer state of the native step, in torch.optim.Adam's own state_dict layout ------------------------------
    # The reference checkpoints through Lightning's ModelCheckpoint (volumetric/base_trainer.py:224-225), which stores
    # ``optimizer.state_dict()`` next to the module's ``state_dict``.  fit_step keeps Adam's moments in the flat store, outside any
    # torch optimizer, so the module exports / imports them in exactly that layout: a checkpoint written after native steps
    # resumes under ``configure_optimizers()``'s torch Adam and vice versa, and ``state_dict()`` keeps the reference's keys only.
    def optimizer_state_dict(self, betas=(0.9, 0.999), eps=1e-8):
        st = self.unet.engine().store
        params = list(self.parameters())
        state = {}
        if st is not None and st.step > 0:
            for i, p in enumerate(params):
                o, n = st.off(p), p.numel()
                state[i] = {"step": torch.tensor(float(st.step)), "exp_avg": st.adam_m[o:o + n].view(p.shape).clone(),
                            "exp_avg_sq": st.adam_v[o:o + n].view(p.shape).clone()}
        group = {"lr": self.hparams.lr, "betas": tuple(betas), "eps": eps, "weight_decay": 0, "amsgrad": False,
                 "params": list(range(len(params)))}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, sd):
        eng = self.unet.engine()
        st = eng.ensure(self.device)
        params = list(self.parameters())
        state = sd["state"]
        if not state:
            st.adam_m = st.adam_v = None
            st.step = 0
            return
        steps = {int(float(v["step"])) for v in state.values()}
        if len(state) != len(params) or len(steps) != 1:
            raise ValueError("optimizer state does not cover every parameter with one common step count")
        st.ensure_adam_state()
        with torch.no_grad():
            for i, p in enumerate(params):
                o, n = st.off(p), p.numel()
                e = state[i] if i in state else state[str(i)]
                st.adam_m[o:o + n].copy_(e["exp_avg"].reshape(-1))
                st.adam_v[o:o + n].copy_(e["exp_avg_sq"].reshape(-1))
        st.step = steps.pop()

    def on_save_checkpoint(self, checkpoint):
        """Lightning hook: the native step's Adam state rides in the checkpoint under its own key"""
        checkpoint["ctseg_native_adam"] = self.optimizer_state_dict()

    def on_load_checkpoint(self, checkpoint):
        if checkpoint.get("ctseg_native_adam", {}).get("state"):
            self.load_optimizer_state_dict(checkpoint["ctseg_native_adam"])

    def checkpoint(self):
        """what Lightning's ModelCheckpoint would write for this module (weights + optimizer state + hparams)"""
        ck = {"state_dict": self.state_dict(), "hyper_parameters": dict(self.hparams),
              "optimizer_states": [self.optimizer_state_dict()]}
        return ck

    def load_checkpoint(self, ck):
        self.load_state_dict(ck["state_dict"])
        if ck.get("optimizer_states"):
            self.load_optimizer_state_dict(ck["optimizer_states"][0])

    @staticmethod
    def add_model_specific_args(parent_parser):
        """Same flags and defaults as reference :134-182."""
        parser = ArgumentParser(parents=[parent_parser], add_help=False)
        parser.add_argument("--batch_size", type=int, default=1, help="Volumes per optimizer step")
        parser.add_argument("--transform_degree", type=int, default=0,
                            help="Augmentation preset index (0 = none)")
        parser.add_argument("--filters", nargs=5, type=int, default=[64, 128, 256, 512, 1024],
                            help="Channel widths of the five encoder levels")
        parser.add_argument("--use_res_units", action="store_true", default=False, help="Build the U-Net from residual units")
        parser.add_argument("--downsample", action="store_true", default=False,
                            help="Reduce a 3-channel input to 1 channel with a 1x1 convolution before the U-Net")
        parser.add_argument("--lr", type=float, default=1e-3, help="Adam step size")
        parser.add_argument("--loss_fx", nargs="+", type=str, default="CrossEntropy", help="One or more loss names, summed")
        parser.add_argument("--exclude_missing", action="store_true", default=False,
                            help="Weight per-class loss terms by annotation availability (AnatomyNet)")
        return parser
