"""GPU: the persistent kernels with SEVERAL tiles per workgroup at small shapes.

Every persistent launcher sizes its grid as min(tiles, CTSEG_NUM_CU * k), so at the op-level test shapes each workgroup runs ONE
tile and nothing between two tiles is exercised: the prefetch of tile t+1 (halo DMA, weight stages issued during the last chunk),
the store of tile t-1 from the software-pipelined epilogue, ring-slot rotation across an epilogue, the counted ``vmcnt`` with
epilogue loads / stores in flight, the statistics flush when the sample changes inside one workgroup.  Round 3 shipped an LDS
overwrite race of exactly that class which only the 2x512x512x48 tests could see.  ``CTSEG_MAX_WG=n`` (test-only, read by
``ctseg::persistent_grid`` in every launcher and sizing query) caps the grid: with 1 or 3 workgroups the same small shapes walk
3 ... 100 tiles per workgroup including a ragged last one and a sample boundary.  Oracle: torch CPU (op level), and the same network
without the cap (network level: same values, another order of the fp32 partial sums).  Run once; coverage, not a stress loop."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from capstone_amd._native import BF16, F32  # noqa: E402
from helpers import rel_err, run_conv_module  # noqa: E402

DEV = "cuda:0"

# (family the forward / input-gradient / weight-gradient passes land on, kind, cin, cout, (N, X, Y, Z))
OP_CASES = [
    ("halo_x 64B voxels + wgrad_halo<64,64>", "conv", 32, 32, (2, 12, 16, 24)),
    ("halo_x 32B voxels, ragged", "conv", 16, 16, (1, 9, 11, 13)),
    ("head: 16->10 halo_x + wgrad_head2, two samples", "conv", 16, 10, (2, 8, 12, 16)),
    ("up_halo 64->10 + wgrad_up + down_halo dgrad", "convT", 64, 10, (2, 8, 8, 8)),
    ("up_halo as the dgrad of a stride-2 conv", "conv_s2", 16, 64, (2, 8, 8, 16)),
    ("halo_sw 64->64", "conv", 64, 64, (2, 16, 16, 12)),
    ("halo_sw 64->64 ragged", "conv", 64, 64, (1, 9, 20, 13)),
    ("halo_sw 8-class 128->32", "convT", 128, 32, (2, 6, 20, 12)),
    ("down_r 32->128 s2 + its 8-class dgrad", "conv_s2", 32, 128, (2, 32, 32, 16)),
    ("down_r ragged", "conv_s2", 32, 128, (1, 18, 40, 24)),
    ("up8 384->64", "convT", 384, 64, (2, 10, 18, 12)),
    ("up8 as the dgrad of 64->256 s2", "conv_s2", 64, 256, (1, 36, 40, 12)),
    ("down_halo 16->64 s2", "conv_s2", 16, 64, (1, 32, 32, 16)),
    ("down_halo as the dgrad of 64->16 convT", "convT", 64, 16, (2, 16, 16, 8)),
    ("stem 1->32 s2", "conv_s2", 1, 32, (2, 8, 16, 16)),
    ("stem 1->64 s2 ragged", "conv_s2", 1, 64, (1, 10, 12, 8)),
]


def _module(kind, cin, cout):
    if kind == "convT":
        return torch.nn.ConvTranspose3d(cin, cout, 3, 2, 1, output_padding=1)
    return torch.nn.Conv3d(cin, cout, 3, 2 if kind == "conv_s2" else 1, 1)


@pytest.mark.parametrize("max_wg", ["1", "3"])
@pytest.mark.parametrize("family,kind,cin,cout,shape", OP_CASES, ids=[c[0] for c in OP_CASES])
def test_op_level_passes_with_capped_grids(monkeypatch, max_wg, family, kind, cin, cout, shape):
    monkeypatch.setenv("CTSEG_MAX_WG", max_wg)
    torch.manual_seed(cin * 7 + cout + shape[2])
    mod = _module(kind, cin, cout)
    x = torch.randn(shape[0], cin, *shape[1:])
    xr = x.clone().requires_grad_(cin > 1)
    y = mod(xr)
    gy = torch.randn_like(y)
    y.backward(gy)
    yy, gx, gw, gb = run_conv_module(mod, x, gy, BF16, DEV)
    assert rel_err(yy, y.detach()) < 2.5e-2, "forward"
    if cin > 1:
        assert rel_err(gx, xr.grad) < 2.5e-2, "input gradient"
    assert rel_err(gw, mod.weight.grad) < 2.5e-2, "weight gradient"
    assert rel_err(gb, mod.bias.grad) < 2.5e-2, "bias gradient"


@pytest.mark.parametrize("max_wg", ["1", "3"])
@pytest.mark.parametrize("cin,cout,shape", [(32, 32, (2, 12, 16, 24)), (16, 10, (2, 8, 12, 16))])
def test_fp32_halo_kernel_with_capped_grids(monkeypatch, max_wg, cin, cout, shape):
    """fp32 storage keeps the register-staged conv_halo_kernel (the 16-bit plans take conv_halo_x first)"""
    monkeypatch.setenv("CTSEG_MAX_WG", max_wg)
    torch.manual_seed(cin + cout)
    mod = torch.nn.Conv3d(cin, cout, 3, 1, 1)
    x = torch.randn(shape[0], cin, *shape[1:])
    xr = x.clone().requires_grad_(True)
    y = mod(xr)
    gy = torch.randn_like(y)
    y.backward(gy)
    yy, gx, gw, gb = run_conv_module(mod, x, gy, F32, DEV)
    assert rel_err(yy, y.detach()) < 2e-5 and rel_err(gx, xr.grad) < 2e-5
    assert rel_err(gw, mod.weight.grad) < 2e-5 and rel_err(gb, mod.bias.grad) < 2e-5


NET_CASES = [
    # (batch shape, filters, caps): statistics epilogues (forward + backward), the fused logits conv + cross-entropy, the
    # DMA-staged addends, dY formed on load — everything the op-level harness above does not reach
    ((2, 32, 48, 16), [16, 32, 64], ("1", "3")),
    ((3, 20, 24, 12), [8, 16], ("3",)),                       # three samples: sample changes inside a workgroup
    ((2, 64, 64, 32), [32, 64, 128, 256], ("3", "7")),        # the benchmark's channels
    ((2, 40, 72, 48), [32, 64, 128], ("3",)),                 # streamed-weight / stride-2 halo producers with backward statistics
    ((1, 36, 44, 20), [32, 64], ("2",)),                      # ragged tiles on every axis
]


def _step(filters, batch, keep_logits, loss_fx=("CrossEntropy",)):
    from capstone_amd.volumetric.base_trainer import BaseUNet3D
    torch.manual_seed(31)
    m = BaseUNet3D(filters=list(filters), loss_fx=list(loss_fx), precision="bf16").to(DEV)
    loss = float(m.fit_step(batch, keep_logits=keep_logits))
    eng = m.unet.engine()
    le = eng.last_plan._ctseg_loss
    torch.cuda.synchronize()
    logits = eng.logits_view().clone() if keep_logits else None
    return loss, eng.store.flat_g.clone(), le.cnt.clone(), logits, float(m.logged["Mean Dice Score (train)"])


@pytest.mark.parametrize("keep_logits", [False, True], ids=["fused_head", "two_pass_head"])
@pytest.mark.parametrize("shape,filters,caps", NET_CASES, ids=["x".join(map(str, c[0])) for c in NET_CASES])
def test_training_step_with_capped_grids_equals_the_uncapped_step(monkeypatch, shape, filters, caps, keep_logits):
    B, H, W, D = shape
    g = torch.Generator().manual_seed(47)
    images = torch.randn(B, 1, H, W, D, generator=g).to(DEV)
    masks = (torch.rand(B, 9, H, W, D, generator=g) < 0.08).to(torch.uint8).to(DEV)
    batch = (images, masks, torch.ones(B, 9, dtype=torch.float64).to(DEV))
    monkeypatch.delenv("CTSEG_MAX_WG", raising=False)
    ref = _step(filters, batch, keep_logits)
    for cap in caps:
        monkeypatch.setenv("CTSEG_MAX_WG", cap)
        got = _step(filters, batch, keep_logits)
        monkeypatch.delenv("CTSEG_MAX_WG")
        # same values, another order of the fp32 partial sums (InstanceNorm statistics per workgroup slot, weight-gradient slabs):
        # a statistic that differs in its last bits re-rounds a few bf16 activations downstream
        assert abs(got[0] - ref[0]) < 2e-3 * max(1.0, abs(ref[0])), (cap, got[0], ref[0])
        a, b = got[1].double(), ref[1].double()
        assert torch.isfinite(a).all()
        cos = float(torch.dot(a, b) / (a.norm() * b.norm()))
        assert cos > 0.9999, (cap, cos)
        flips = int((got[2] - ref[2]).abs().sum())                  # Dice counts: integer, a handful of argmax near-ties may flip
        assert flips <= max(8, ref[2][:, 1].sum().item() * 2e-3), (cap, flips)
        assert abs(got[4] - ref[4]) < 2e-3
        if keep_logits:
            scale = float(ref[3].abs().max())
            assert float((got[3] - ref[3]).abs().max()) < 3e-2 * scale, cap


def test_dice_focal_drop_in_step_with_capped_grids(monkeypatch):
    """the non-fused route (training_step -> loss.backward() with Dice + Focal: plan._UNetFn) under the cap"""
    from capstone_amd.volumetric.base_trainer import BaseUNet3D
    g = torch.Generator().manual_seed(53)
    images = torch.randn(2, 1, 32, 48, 16, generator=g).to(DEV)
    masks = (torch.rand(2, 9, 32, 48, 16, generator=g) < 0.08).to(torch.uint8).to(DEV)
    batch = (images, masks, torch.ones(2, 9, dtype=torch.float64).to(DEV))
    grads = {}
    for cap in (None, "3"):
        if cap is None:
            monkeypatch.delenv("CTSEG_MAX_WG", raising=False)
        else:
            monkeypatch.setenv("CTSEG_MAX_WG", cap)
        torch.manual_seed(5)
        m = BaseUNet3D(filters=[16, 32, 64], loss_fx=["Dice", "Focal"], precision="bf16").to(DEV)
        loss = m.training_step(batch, 0)
        loss.backward()
        torch.cuda.synchronize()
        grads[cap] = (float(loss.detach()), m.unet.engine().store.flat_g.clone().double())
    monkeypatch.delenv("CTSEG_MAX_WG", raising=False)
    assert abs(grads["3"][0] - grads[None][0]) < 2e-3 * abs(grads[None][0])
    a, b = grads["3"][1], grads[None][1]
    assert float(torch.dot(a, b) / (a.norm() * b.norm())) > 0.9999


# ----------------------------------------------------------------------------------------------------------------------
# Round 4: the slab reduces of a backward pass as ONE launch per gradient chunk (ctseg_conv_wgrad_reduce_batch)
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape,filters,precision", [((2, 32, 48, 16), [16, 32, 64], "bf16"), ((2, 64, 64, 32), [32, 64, 128, 256], "bf16"),
                                                     ((1, 24, 40, 16), [8, 16, 32], "fp32")])
def test_batched_slab_reduce_is_bit_identical_to_the_per_pass_reduces(monkeypatch, shape, filters, precision):
    """same partition of the slabs, same fixed-order combine -> the SAME flat gradient bit for bit, from 3 launches instead of one
    per weight-gradient pass; the readiness marks still give the data-parallel exchange a chunk to send mid-backward"""
    from capstone_amd import distributed as cdist
    from capstone_amd.volumetric.base_trainer import BaseUNet3D
    B, H, W, D = shape
    g = torch.Generator().manual_seed(61)
    images = torch.randn(B, 1, H, W, D, generator=g).to(DEV)
    masks = (torch.rand(B, 9, H, W, D, generator=g) < 0.08).to(torch.uint8).to(DEV)
    batch = (images, masks, torch.ones(B, 9, dtype=torch.float64).to(DEV))
    out = {}
    for on in ("0", "1"):
        monkeypatch.setenv("CTSEG_REDUCE_BATCH", on)
        torch.manual_seed(19)
        m = BaseUNet3D(filters=list(filters), loss_fx=["CrossEntropy"], precision=precision).to(DEV)
        loss = float(m.fit_step(batch, keep_logits=False))
        eng = m.unet.engine()
        plan = eng.last_plan
        names = [nm for nm, *_ in plan.bwd]
        torch.cuda.synchronize()
        st = eng.store
        sizes = {st.off(p): p.numel() for p in st.params}
        out[on] = (loss, st.flat_g.clone(), names.count("ctseg_conv_wgrad_reduce"), names.count("ctseg_conv_wgrad_reduce_batch"),
                   cdist.split_points(plan.ready_marks, sizes, st.n), st.flat_p.clone())
    a, b = out["0"], out["1"]
    assert a[3] == 0 and a[2] >= 6
    assert b[2] == 0 and 1 <= b[3] <= 3, b[2:4]
    assert a[0] == b[0]
    assert torch.equal(a[1], b[1]), float((a[1] - b[1]).abs().max())
    assert torch.equal(a[5], b[5])                       # ... and so is the Adam update
    assert len(b[4]) >= 1 and all(0 < end < a[1].numel() for _, end in b[4])
