"""Ring-pipelined weight-gradient kernel (csrc/conv_wgrad_ring.hip) against the generic kernel it replaces and against autograd.

The pass is autograd's Conv3d / ConvTranspose3d weight + bias backward inside the reference's training step
(/root/reference/capstone/volumetric/base_trainer.py:80-82 -> loss.backward()) for the many-channel layers MONAI's UNet builds at :65-72.
Both kernels sum bf16 products in fp32, in a different order: they agree to fp32 rounding of a sum of `rows` terms, and both agree with
torch's fp32 autograd on the SAME bf16-rounded operands to 2e-3 of the largest gradient entry (the products are exact in fp32; only the
summation order differs).  Shapes: every tile (256x128, 512x64; 256x256 when enabled), stride 1 and 2, the
transposed roles, row counts that are not a multiple of a 32-voxel stage, fewer rows than a workgroup's minimum of stages, several
splits with an empty last split, a K extent that ends exactly on a tile edge (the bias row then has a tile of its own).
"""
import ctypes
import os

import numpy as np
import pytest
import torch

from capstone_amd import _native as nat
from capstone_amd.engine import GemmLayer, rup

from helpers import MiniPlan, to_cl, rel_err

pytestmark = pytest.mark.gpu

CASES = [
    # name, module ctor, input spatial dims (N, X, Y, Z)
    ("64->64 s1 (512x64 tile), ragged rows", lambda: torch.nn.Conv3d(64, 64, 3, 1, 1), (2, 7, 10, 6)),
    ("32->128 s2 (256x128 tile, K = 865 of 1024)", lambda: torch.nn.Conv3d(32, 128, 3, 2, 1), (2, 12, 16, 8)),
    ("128->128 s1 (bias row in a tile of its own: K = 3456 = 13.5 x 256)", lambda: torch.nn.Conv3d(128, 128, 3, 1, 1), (1, 6, 8, 6)),
    ("64->256 s1 (256x256 tile)", lambda: torch.nn.Conv3d(64, 256, 3, 1, 1), (2, 6, 6, 6)),
    ("256->256 s1 (256x256 tile; K = 6912 = 27 x 256: bias-only last tile)", lambda: torch.nn.Conv3d(256, 256, 3, 1, 1), (1, 4, 6, 6)),
    ("128->256 s1 (256x256 tile, bias row inside the last tile), two samples", lambda: torch.nn.Conv3d(128, 256, 3, 1, 1), (2, 5, 6, 7)),
    ("64->384 s2 (three column tiles)", lambda: torch.nn.Conv3d(64, 384, 3, 2, 1), (1, 8, 8, 12)),
    ("transposed 128->32 s2 (roles swapped: gathered = dOut)", lambda: torch.nn.ConvTranspose3d(128, 32, 3, 2, 1, output_padding=1), (2, 6, 6, 4)),
    ("transposed 256->64 s2", lambda: torch.nn.ConvTranspose3d(256, 64, 3, 2, 1, output_padding=1), (1, 4, 6, 6)),
    ("1x1x1 128->64 (one tap)", lambda: torch.nn.Conv3d(128, 64, 1, 1, 0), (2, 9, 5, 7)),
]


def _wgrad(mod, x, gy, ring, splits_env=None):
    """weight / bias gradient of one module through GemmLayer.emit_wgrad; returns (gw, gb, used_ring, splits)"""
    os.environ["CTSEG_WGRAD_RING"] = "1" if ring else "0"
    try:
        dt, device = nat.BF16, "cuda"
        transposed = isinstance(mod, torch.nn.ConvTranspose3d)
        plan = MiniPlan([mod.weight, mod.bias], device, dt, 3)
        cin = mod.in_channels
        xa = to_cl(x, dt, device)
        layer = GemmLayer(plan, "t", transposed, mod.kernel_size[0], mod.stride[0], cin, [(mod.weight, mod.bias, mod.out_channels)], rup(cin, 8),
                          need_dgrad=True)
        plan.packer.finalize()
        ga = to_cl(gy, dt, device)
        layer.emit_wgrad(xa, ga)
        desc = [a[0] for n, f, a in plan.prog if n == "ctseg_conv_wgrad"][0]
        per_cu = ctypes.c_int32(0)
        nat.lib().ctseg_conv_wgrad_wgs_per_slab(ctypes.byref(desc), ctypes.byref(per_cu), None)
        plan.run()
        torch.cuda.synchronize()
        gw = plan.store.grad_view(mod.weight).cpu().clone()
        gb = plan.store.grad_view(mod.bias).cpu().clone() if not transposed else None
        return gw, gb, per_cu.value == 1, desc.splits
    finally:
        os.environ.pop("CTSEG_WGRAD_RING", None)


@pytest.mark.parametrize("name,ctor,dims", CASES, ids=[c[0] for c in CASES])
def test_ring_kernel_equals_the_generic_kernel_and_autograd(name, ctor, dims):
    torch.manual_seed(hash(name) % 1000)
    mod = ctor()
    N, X, Y, Z = dims
    x = torch.randn(N, mod.in_channels, X, Y, Z).bfloat16().float()
    xr = x.clone().requires_grad_(False)
    y = mod(xr)
    gy = torch.randn_like(y).bfloat16().float()
    # autograd on the bf16-rounded operands
    mod.zero_grad()
    mod(xr).backward(gy)
    ref_w, ref_b = mod.weight.grad.clone(), mod.bias.grad.clone()

    gw1, gb1, used, splits = _wgrad(mod, x, gy, ring=True)
    assert used, "the ring kernel did not take this layer (ctseg_conv_wgrad_wgs_per_slab)"
    gw0, gb0, used0, _ = _wgrad(mod, x, gy, ring=False)
    assert not used0
    assert rel_err(gw1, gw0) < 2e-5, (name, rel_err(gw1, gw0))
    assert rel_err(gw1, ref_w) < 2e-3, (name, rel_err(gw1, ref_w))       # bf16 operands, fp32 sums: only the order differs
    if gb1 is not None:
        assert rel_err(gb1, gb0) < 2e-5 and rel_err(gb1, ref_b) < 2e-3, (name, rel_err(gb1, gb0), rel_err(gb1, ref_b))


def test_ring_kernel_is_deterministic_and_takes_any_split_count():
    """a run is bit-reproducible, and forcing odd split counts (last split short or empty) changes only the summation order"""
    torch.manual_seed(3)
    mod = torch.nn.Conv3d(64, 128, 3, 1, 1)
    x = torch.randn(2, 64, 12, 12, 10).bfloat16().float()
    gy = torch.randn(2, 128, 12, 12, 10).bfloat16().float()
    a = _wgrad(mod, x, gy, ring=True)
    b = _wgrad(mod, x, gy, ring=True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    outs = []
    for target in ("20", "50", "4096"):         # the generic split rule: 1 440 rows in 1, 2 (the second one short) and 3 ranges
        os.environ["CTSEG_WGRAD_TARGET_WGS"] = target
        try:
            outs.append(_wgrad(mod, x, gy, ring=True))
        finally:
            os.environ.pop("CTSEG_WGRAD_TARGET_WGS", None)
    assert len({o[3] for o in outs}) > 1, [o[3] for o in outs]
    for o in outs:
        assert o[2]
        assert rel_err(o[0], a[0]) < 2e-5 and rel_err(o[1], a[1]) < 2e-5
