"""Mirror of reference capstone/models/__init__.py:1-3."""
from .losses import MultipleLossWrapper
from .metrics import DiceMetricWrapper
from .unet import UNet

__all__ = ["MultipleLossWrapper", "DiceMetricWrapper", "UNet"]
