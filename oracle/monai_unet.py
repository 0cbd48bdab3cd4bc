"""Oracle: pure-``torch.nn`` CPU restatement of MONAI 0.3's ``UNet``.

TEST INFRASTRUCTURE (see oracle/__init__.py). **Parity unpinned**: MONAI 0.3 is a
third-party dependency of the reference (``from monai.networks.nets import UNet``,
reference capstone/models/__init__.py:3; version pinned in prose at README.md:39)
that is not vendored and not installable here. The topology below restates the
published MONAI 0.3.0 ``nets/unet.py`` + ``blocks/convolutions.py`` +
``layers/simplelayers.py::SkipConnection`` as summarised in SURVEY.md §3.2; the
module tree is corroborated by the reference's own indexing
``model.unet.model[2][1].conv.unit0.conv`` (capstone/interpretability.py:88).

Call sites this mirrors: capstone/volumetric/base_trainer.py:65-72 (3-D),
capstone/training/base_trainer.py:72-79 (2-D).
"""
from typing import Sequence

import torch
import torch.nn as nn

_CONV = {2: nn.Conv2d, 3: nn.Conv3d}
_CONVT = {2: nn.ConvTranspose2d, 3: nn.ConvTranspose3d}
_INORM = {2: nn.InstanceNorm2d, 3: nn.InstanceNorm3d}


def _check_act_norm(act, norm, dropout):
    if str(act).upper() != "PRELU" or str(norm).upper() != "INSTANCE" or dropout:
        raise NotImplementedError("oracle restates only act=PRELU, norm=INSTANCE, dropout=0")


class Convolution(nn.Sequential):
    """conv (or transposed conv) [-> InstanceNorm -> PReLU]; children named conv/norm/act."""

    def __init__(self, dims, cin, cout, strides=1, kernel_size=3, conv_only=False, is_transposed=False):
        super().__init__()
        pad = (kernel_size - 1) // 2
        if is_transposed:
            conv = _CONVT[dims](cin, cout, kernel_size, stride=strides, padding=pad, output_padding=strides - 1)
        else:
            conv = _CONV[dims](cin, cout, kernel_size, stride=strides, padding=pad)
        self.add_module("conv", conv)
        if not conv_only:
            self.add_module("norm", _INORM[dims](cout))
            self.add_module("act", nn.PReLU())


class ResidualUnit(nn.Module):
    """``conv(x) + residual(x)`` with ``subunits`` Convolution blocks (first carries the stride)."""

    def __init__(self, dims, cin, cout, strides=1, kernel_size=3, subunits=2, last_conv_only=False):
        super().__init__()
        self.conv = nn.Sequential()
        self.residual = nn.Identity()
        subunits = max(1, subunits)
        c, s = cin, strides
        for su in range(subunits):
            only = last_conv_only and su == subunits - 1
            self.conv.add_module(f"unit{su:d}", Convolution(dims, c, cout, s, kernel_size, conv_only=only))
            c, s = cout, 1
        if strides != 1 or cin != cout:
            if strides == 1:  # channel change only: 1x1 kernel, no padding
                self.residual = _CONV[dims](cin, cout, 1, strides, 0)
            else:
                self.residual = _CONV[dims](cin, cout, kernel_size, strides, (kernel_size - 1) // 2)

    def forward(self, x):
        res = self.residual(x)
        return self.conv(x) + res


class SkipConnection(nn.Module):
    def __init__(self, submodule):
        super().__init__()
        self.submodule = submodule

    def forward(self, x):
        return torch.cat([x, self.submodule(x)], 1)


class UNet(nn.Module):
    def __init__(self, dimensions: int, in_channels: int, out_channels: int, channels: Sequence[int],
                 strides: Sequence[int], kernel_size=3, up_kernel_size=3, num_res_units: int = 0,
                 act="PRELU", norm="INSTANCE", dropout=0):
        super().__init__()
        _check_act_norm(act, norm, dropout)
        self.dimensions, self.in_channels, self.out_channels = dimensions, in_channels, out_channels
        self.channels, self.strides = list(channels), list(strides)
        self.kernel_size, self.up_kernel_size, self.num_res_units = kernel_size, up_kernel_size, num_res_units

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

    def forward(self, x):
        return self.model(x)
