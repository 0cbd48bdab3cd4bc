"""capstone_amd — MI355X-native drop-in for the reference's ``capstone`` hot path.

Module names mirror the reference package (``capstone.models``, ``capstone.volumetric``,
``capstone.training``) so that ``from capstone_amd.models import UNet`` replaces
``from capstone.models import UNet``.  Everything arithmetic runs in libctseg_hip.so
(hand-written HIP for gfx950, include/ctseg_hip.h); PyTorch supplies device memory, streams and
``torch.distributed`` only.
"""
from . import _native  # noqa: F401

STRUCTURES = [  # capstone/utils/miccai.py:14-24 — class order the whole reference depends on
    "BrainStem", "Chiasm", "Mandible", "OpticNerve_L", "OpticNerve_R",
    "Parotid_L", "Parotid_R", "Submandibular_L", "Submandibular_R",
]
__all__ = ["STRUCTURES"]
