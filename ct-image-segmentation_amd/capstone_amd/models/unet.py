"""MI355X-native ``UNet`` with MONAI 0.3's constructor signature, module tree and state_dict keys.

Drop-in for ``from capstone.models import UNet`` (reference capstone/models/__init__.py:3 re-exports
``monai.networks.nets.UNet``; constructed at capstone/volumetric/base_trainer.py:65-72 and
capstone/training/base_trainer.py:72-79).  The submodules are PARAMETER CONTAINERS: they give the
same ``state_dict`` keys (``model.0.conv.unit0.conv.weight`` ...), the same default initialisation /
RNG consumption order as torch+MONAI, and the indexable tree the reference relies on
(``model.unet.model[2][1].conv.unit0.conv``, capstone/interpretability.py:88) — but none of them
computes anything.  ``forward`` hands the whole network to the HIP engine (capstone_amd.plan); there
is no eager / CPU path and a CPU tensor raises.

Extra keyword (superset of MONAI's signature): ``precision`` = "fp32" (default: fp32 storage,
v_mfma_f32_16x16x4_f32, reference numerics) or "bf16" (bf16 storage + MFMA, fp32 accumulate, fp32
logits) — the reference reaches reduced precision only through Lightning's --precision flag.
"""
from typing import Sequence

import torch
import torch.nn as nn

from .. import _native as nat

_CONV = {2: nn.Conv2d, 3: nn.Conv3d}
_CONVT = {2: nn.ConvTranspose2d, 3: nn.ConvTranspose3d}
_INORM = {2: nn.InstanceNorm2d, 3: nn.InstanceNorm3d}


def _no_eager(self, *a, **k):
    raise nat.NativeError("this module is a parameter container of the MI355X UNet; call the UNet itself "
                          "(there is no eager fallback)")


class Convolution(nn.Sequential):
    """conv | transposed conv [-> InstanceNorm -> PReLU]  (children: conv, norm, act)"""

    def __init__(self, dims, cin, cout, strides=1, kernel_size=3, conv_only=False, is_transposed=False):
        super().__init__()
        pad = (kernel_size - 1) // 2
        if is_transposed:
            conv = _CONVT[dims](cin, cout, kernel_size, stride=strides, padding=pad, output_padding=strides - 1)
        else:
            conv = _CONV[dims](cin, cout, kernel_size, stride=strides, padding=pad)
        self.add_module("conv", conv)
        self.conv_only, self.is_transposed = conv_only, is_transposed
        self.cin, self.cout, self.strides, self.kernel_size = cin, cout, strides, kernel_size
        if not conv_only:
            self.add_module("norm", _INORM[dims](cout))
            self.add_module("act", nn.PReLU())

    forward = _no_eager


class ResidualUnit(nn.Module):
    def __init__(self, dims, cin, cout, strides=1, kernel_size=3, subunits=2, last_conv_only=False):
        super().__init__()
        self.conv = nn.Sequential()
        self.residual = nn.Identity()
        self.cin, self.cout, self.strides, self.kernel_size = cin, cout, strides, kernel_size
        subunits = max(1, subunits)
        c, s = cin, strides
        for su in range(subunits):
            only = last_conv_only and su == subunits - 1
            self.conv.add_module(f"unit{su:d}", Convolution(dims, c, cout, s, kernel_size, conv_only=only))
            c, s = cout, 1
        if strides != 1 or cin != cout:
            if strides == 1:
                self.residual = _CONV[dims](cin, cout, 1, strides, 0)
            else:
                self.residual = _CONV[dims](cin, cout, kernel_size, strides, (kernel_size - 1) // 2)

    forward = _no_eager


class SkipConnection(nn.Module):
    def __init__(self, submodule):
        super().__init__()
        self.submodule = submodule

    forward = _no_eager


class UNet(nn.Module):
    def __init__(self, dimensions: int, in_channels: int, out_channels: int, channels: Sequence[int],
                 strides: Sequence[int], kernel_size=3, up_kernel_size=3, num_res_units: int = 0,
                 act="PRELU", norm="INSTANCE", dropout=0, *, precision: str = "fp32"):
        super().__init__()
        if str(act).upper() != "PRELU" or str(norm).upper() != "INSTANCE":
            raise NotImplementedError("MI355X UNet implements the reference's act='PRELU', norm='INSTANCE' only")
        if dropout:
            raise NotImplementedError("dropout > 0 is never used by the reference and is not implemented")
        if kernel_size != 3 or up_kernel_size != 3:
            raise NotImplementedError("kernel_size / up_kernel_size other than 3 are not implemented")
        if dimensions not in (2, 3):
            raise ValueError("dimensions must be 2 or 3")
        if precision not in ("fp32", "bf16", "fp16"):
            raise ValueError("precision must be 'fp32', 'bf16' or 'fp16' (inference only)")
        assert len(channels) >= 2 and len(strides) >= len(channels) - 1
        self.dimensions, self.in_channels, self.out_channels = dimensions, in_channels, out_channels
        self.channels, self.strides = list(channels), list(strides)
        self.kernel_size, self.up_kernel_size, self.num_res_units = kernel_size, up_kernel_size, num_res_units
        self.precision = precision

        def block(inc, outc, chans, strs, is_top):
            c, s = chans[0], strs[0]
            if len(chans) > 2:
                sub, upc = block(c, c, chans[1:], strs[1:], False), 2 * c
            else:
                sub, upc = self._down(c, chans[1], 1), c + chans[1]
            down = self._down(inc, c, s)
            up = self._up(upc, outc, s, is_top)
            return nn.Sequential(down, SkipConnection(sub), up)

        self.model = block(in_channels, out_channels, self.channels, self.strides, True)
        self._engine = None

    def _down(self, cin, cout, s):
        if self.num_res_units > 0:
            return ResidualUnit(self.dimensions, cin, cout, s, self.kernel_size, self.num_res_units)
        return Convolution(self.dimensions, cin, cout, s, self.kernel_size)

    def _up(self, cin, cout, s, is_top):
        conv = Convolution(self.dimensions, cin, cout, s, self.up_kernel_size,
                           conv_only=is_top and self.num_res_units == 0, is_transposed=True)
        if self.num_res_units > 0:
            ru = ResidualUnit(self.dimensions, cout, cout, 1, self.kernel_size, 1, last_conv_only=is_top)
            conv = nn.Sequential(conv, ru)
        return conv

    # ---- engine plumbing --------------------------------------------------------------------------
    def engine(self):
        from ..plan import Engine
        if self._engine is None:
            self._engine = Engine(self)
        return self._engine

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(B, Cin, *spatial) -> fp32 logits (B, Cout, *spatial) (a channels-last strided view)."""
        nat.require_gpu(x, "UNet.forward")
        return self.engine().forward_autograd(x)
