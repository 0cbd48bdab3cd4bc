"""Drop-in surface of reference capstone/training/base_trainer.py:22-148 (``BaseUNet2D``) on the MI355X engine.

Row a14 of SURVEY.md §8: the 2-D model is BASELINE.json's configs[0] *plumbing* case — same constructor, hparams,
``forward`` / ``training_step`` / ``validation_step`` / ``test_step`` / ``_shared_step`` / ``configure_optimizers`` (Adam +
ReduceLROnPlateau(mode="max", factor=0.5, threshold=0.01) monitoring "Mean Dice Score (val)").  A 2-D U-Net is the 3-D
engine with Z = 1 (9-tap convolutions, 4-class transposed convs); no 2-D-specific kernel exists or is needed.

``--downsample`` (reference :53,81-85: a trainable 3->1 ``conv1x1`` in front of the U-Net) is carried: the channel mix is three
elementwise multiply-adds on the input image and the engine records an input-gradient pass for the stem when its input requires grad.
Not carried over (outside the hot path): the Boundary loss (CPU distance maps), mixup and the W&B logger patch.  They
raise instead of silently doing something else.
"""
from argparse import ArgumentParser
from typing import List

import torch

from .. import STRUCTURES
from ..models import DiceMetricWrapper, MultipleLossWrapper, UNet
from ..volumetric.base_trainer import _Base, _DataParallelSurface, _precision, pl
from .utils import _squash_masks, _squash_predictions


class BaseUNet2D(_DataParallelSurface, _Base):
    def __init__(self, filters: List = [64, 128, 256, 512, 1024], use_res_units: bool = False, downsample: bool = False,
                 lr: float = 1e-3, loss_fx: list = ["Focal", "Dice"], exclude_missing: bool = False, **kwargs) -> None:
        super().__init__()
        assert isinstance(filters, list)
        assert len(filters) == 5, "This module requires a standard 5 block UNet specification"
        assert isinstance(loss_fx, list), "This module expects a list of loss functions"
        loss_fx.sort()
        names = ("batch_size", "transform_degree", "filters", "use_res_units", "downsample", "lr", "loss_fx", "exclude_missing")
        if pl is not None:
            self.save_hyperparameters(*names)
        else:
            self.save_hyperparameters(*names, frame_locals=dict(locals()))
        self._precision = _precision(kwargs)
        # reference :53 builds this 3->1 convolution whatever ``downsample`` says, so its two tensors are in every reference
        # checkpoint: kept as a parameter container for state_dict interchange
        self.conv1x1 = torch.nn.Conv2d(in_channels=3, out_channels=1, kernel_size=1, stride=1)
        self.unet = self._construct_model()
        self.loss_func = MultipleLossWrapper(losses=loss_fx, exclude_missing=exclude_missing)
        self.dice_score = DiceMetricWrapper()

    @property
    def _n_classes(self):
        return len(STRUCTURES) + 1

    def _construct_model(self):
        in_channels = 1 if (self.hparams.downsample or (self.hparams.get("transform_degree") in (0, None))) else 3
        return UNet(dimensions=2, in_channels=in_channels, out_channels=self._n_classes, channels=self.hparams.filters,
                    strides=[2, 2, 2, 2], num_res_units=(2 if self.hparams.use_res_units else 0), precision=self._precision)

    def forward(self, x):
        if self.hparams.downsample:
            # reference :81-85 — a trainable 3 -> 1 channel mix in front of the U-Net.  Three multiply-adds per pixel on the INPUT
            # image: plain elementwise device ops (not a convolution library call); its gradient comes from the U-Net's input-gradient
            # pass, which the engine records when the input requires grad
            w, b = self.conv1x1.weight, self.conv1x1.bias
            x = (x * w.view(1, -1, 1, 1)).sum(dim=1, keepdim=True) + b.view(1, 1, 1, 1)
        return self.unet(x)

    def training_step(self, batch, batch_idx=0):
        return self._shared_step(batch, prefix="train")[-1]

    def validation_step(self, batch, batch_idx=0):
        self._shared_step(batch, prefix="val")

    def test_step(self, batch, batch_idx=0):
        self._shared_step(batch, prefix="test")

    def _shared_step(self, batch, prefix: str):
        images, masks, mask_indicator, *dist_maps = batch
        if dist_maps:
            raise NotImplementedError("distance maps (Boundary loss) are outside the MI355X hot path")
        masks = _squash_masks(masks, self._n_classes, self.device)
        mask_indicator = mask_indicator.type_as(images)
        prediction = self.forward(images)
        prediction._ctseg_plan = self.unet.engine().last_plan if prediction.requires_grad else None
        loss_dict = self.loss_func(input=prediction, target=masks, mask_indicator=mask_indicator)
        total_loss = torch.stack(list(loss_dict.values())).sum()
        for name, loss_value in loss_dict.items():
            self.log(f"{name} Loss ({prefix})", loss_value, on_step=False, on_epoch=True)
        self._log_dice_scores(prediction, masks, mask_indicator, prefix)
        return images, masks, mask_indicator, prediction, total_loss

    def _log_dice_scores(self, prediction, masks, mask_indicator, prefix):
        with torch.no_grad():
            pred = prediction.detach()
            if self.hparams.exclude_missing:      # reference :124-126: no indicator for background
                pred = pred.clone()
                pred[:, 1:] = pred[:, 1:] * mask_indicator[:, :, None, None]
            pred = _squash_predictions(pred)
            dice_mean, dice_per_class = self.dice_score(pred, masks)
            self._keep_dice_counts(self.dice_score.last_counts, prefix)
            for structure, score in zip(STRUCTURES, dice_per_class):
                self.log(f"{structure} Dice ({prefix})", score, on_step=False, on_epoch=True)
            self.log(f"Mean Dice Score ({prefix})", dice_mean, on_step=False, on_epoch=True)

    def configure_optimizers(self):
        from ..optim import Adam            # a torch.optim.Adam whose step() is one launch for the U-Net's flat parameter buffer
        optimizer = Adam(self.parameters(), lr=self.hparams.lr, unet=self.unet)
        scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, mode="max", factor=0.5, threshold=0.01)
        return {"optimizer": optimizer, "lr_scheduler": scheduler, "monitor": "Mean Dice Score (val)"}

    @staticmethod
    def add_model_specific_args(parent_parser):
        """Same flags and defaults as reference :150-210."""
        parser = ArgumentParser(parents=[parent_parser], add_help=False)
        parser.add_argument("--batch_size", type=int, default=128, help="Slices per optimizer step")
        parser.add_argument("--transform_degree", type=int, default=0,
                            help="Augmentation preset index (0 = none)")
        parser.add_argument("--filters", nargs=5, type=int, default=[64, 128, 256, 512, 1024],
                            help="Channel widths of the five encoder levels")
        parser.add_argument("--use_res_units", action="store_true", default=False, help="Build the U-Net from residual units")
        parser.add_argument("--downsample", action="store_true", default=False,
                            help="Reduce a 3-channel input to 1 channel with a 1x1 convolution before the U-Net")
        parser.add_argument("--lr", type=float, default=1e-3, help="Adam step size")
        parser.add_argument("--loss_fx", nargs="+", type=str, default=["Focal", "Dice"], help="One or more loss names, summed")
        parser.add_argument("--exclude_missing", action="store_true", default=False,
                            help="Weight per-class loss terms by annotation availability (AnatomyNet)")
        return parser
