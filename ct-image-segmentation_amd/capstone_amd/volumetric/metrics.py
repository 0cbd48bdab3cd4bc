"""Mirror of reference capstone/volumetric/metrics.py: the 3-D metric wrapper (rank-agnostic here)."""
from ..models.metrics import DiceMetricWrapper


class DiceMetricWrapper3D(DiceMetricWrapper):
    pass
