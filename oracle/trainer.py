"""Oracle: the reference's 3-D training step as a plain ``nn.Module`` (no Lightning).

TEST INFRASTRUCTURE (see oracle/__init__.py). Follows
capstone/volumetric/base_trainer.py:22-132: squash masks (:91-93) -> forward (:97-99)
-> losses (:101-104) -> softmax/argmax Dice on a detached copy (:116-132); Adam at :113-114.
Lightning 1.0's per-batch order (zero_grad -> training_step -> backward -> step) is
reproduced by ``fit_step``. Also the timed ``cpu_baseline`` ("port") of bench.py.
"""
import torch
import torch.nn as nn

from .losses import MultipleLoss
from .metrics import N_CLASSES, DiceMetric, squash_masks, squash_predictions
from .monai_unet import UNet

SEED = 12342  # capstone/volumetric/base_trainer.py:18


class OracleUNet3D(nn.Module):
    def __init__(self, filters=(16, 32, 64, 128, 256), lr=1e-3, loss_fx=("CrossEntropy",),
                 exclude_missing=False, dimensions=3):
        super().__init__()
        self.lr = lr
        self.unet = UNet(dimensions=dimensions, in_channels=1, out_channels=N_CLASSES, channels=list(filters),
                         strides=[2, 2, 2, 2], num_res_units=2)
        self.loss_func = MultipleLoss(sorted(loss_fx), exclude_missing)
        self.dice_score = DiceMetric()
        self.logged = {}

    def forward(self, x):
        return self.unet(x)

    def shared_step(self, batch, is_training=True):
        images, masks, indicator = batch
        labels = squash_masks(masks, N_CLASSES)
        indicator = indicator.type_as(images)
        prefix = "train" if is_training else "val"
        logits = self.forward(images)
        losses = self.loss_func(input=logits, target=labels, mask_indicator=indicator)
        total = torch.stack(list(losses.values())).sum()
        for k, v in losses.items():
            self.logged[f"{k} Loss ({prefix})"] = v.detach()
        with torch.no_grad():
            pred = squash_predictions(logits.detach().clone())
            mean_dice, per_class = self.dice_score(pred, labels)
        self.logged[f"Mean Dice Score ({prefix})"] = mean_dice
        self.logged[f"Dice per class ({prefix})"] = per_class
        return images, labels, indicator, logits, total

    def training_step(self, batch, batch_idx=0):
        return self.shared_step(batch, True)[-1]

    def configure_optimizers(self):
        return torch.optim.Adam(self.parameters(), lr=self.lr)

    def fit_step(self, batch, optimizer):
        optimizer.zero_grad()
        loss = self.training_step(batch)
        loss.backward()
        optimizer.step()
        return loss.detach()
