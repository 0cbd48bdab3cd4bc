"""Mirror of reference capstone/volumetric/losses.py:119-130 (the 3-D loss registry + wrapper)."""
from ..models.losses import LOSSES, WEIGHT, MultipleLossWrapper  # noqa: F401


class MultipleLossWrapper3D(MultipleLossWrapper):
    def __init__(self, losses, exclude_missing=False):
        super(MultipleLossWrapper3D, self).__init__(losses, exclude_missing)
