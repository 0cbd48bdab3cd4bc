"""GPU parity tests (run with -m gpu on a MI355X): the HIP path, called through the C ABI, against
  * torch CPU primitives (op-level fixtures, tests/golden/ops_torch.npz),
  * fixtures produced by the reference's own code (tests/golden/ref_leaf.npz),
  * the oracle restatement of the MONAI-0.3 UNet step (tests/golden/unet_tiny.npz and live oracle runs).
Tolerances: fp32 storage = v_mfma_f32_16x16x4_f32 (an fmaf chain): logits within 1e-3 absolute (north_star),
in practice ~1e-5; integer work (label maps, Dice counts, argmax masks given identical logits) bit-exact;
bf16 storage: relative L-inf error bounds written at each assertion.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from capstone_amd import _native as nat  # noqa: E402
from capstone_amd import segloss  # noqa: E402
from capstone_amd._native import BF16, F32  # noqa: E402
from helpers import MiniPlan, from_cl, rel_err, run_conv_module, to_cl  # noqa: E402

DEV = "cuda:0"


def _mods():
    import torch.nn as nn
    return {
        "conv_k3s1": lambda: nn.Conv3d(8, 12, 3, 1, 1), "conv_k3s2": lambda: nn.Conv3d(8, 16, 3, 2, 1),
        "conv_k3s2_c1": lambda: nn.Conv3d(1, 8, 3, 2, 1), "conv_k1": lambda: nn.Conv3d(16, 24, 1, 1, 0),
        "convT": lambda: nn.ConvTranspose3d(24, 8, 3, 2, 1, output_padding=1),
        "convT_c10": lambda: nn.ConvTranspose3d(16, 10, 3, 2, 1, output_padding=1),
        "conv2d_k3s2": lambda: nn.Conv2d(8, 8, 3, 2, 1),
    }


def test_native_library_is_the_in_tree_build():
    L = nat.lib()
    assert L.ctseg_abi_version() == nat.ABI_VERSION == 3
    assert os.path.realpath(nat.LIB_PATH).startswith(os.path.realpath(os.path.join(os.path.dirname(__file__), "..")))
    assert any("libctseg_hip.so" in line for line in open("/proc/self/maps"))


@pytest.mark.parametrize("dt,tol", [(F32, 2e-5), (BF16, 2.5e-2)])
@pytest.mark.parametrize("tag", list(_mods()))
def test_conv_ops_vs_torch_cpu(golden, tag, dt, tol):
    g = golden("ops_torch.npz")
    mod = _mods()[tag]()
    with torch.no_grad():
        mod.weight.copy_(torch.from_numpy(g[f"{tag}_weight"]))
        mod.bias.copy_(torch.from_numpy(g[f"{tag}_bias"]))
    x, gy = torch.from_numpy(g[f"{tag}_x"]), torch.from_numpy(g[f"{tag}_gy"])
    y, gx, gw, gb = run_conv_module(mod, x, gy, dt, DEV)
    assert rel_err(y, g[f"{tag}_y"]) < tol, "forward"
    if gx is not None:
        assert rel_err(gx, g[f"{tag}_gx"]) < tol, "input gradient"
    assert rel_err(gw, g[f"{tag}_gweight"]) < tol, "weight gradient"
    assert rel_err(gb, g[f"{tag}_gbias"]) < tol, "bias gradient"


@pytest.mark.parametrize("dt,tol", [(F32, 2e-5), (BF16, 2.5e-2)])
@pytest.mark.parametrize("cin,cout,shape", [(32, 32, (2, 12, 16, 24)), (16, 16, (1, 9, 11, 13)), (16, 10, (1, 8, 8, 8)),
                                            (8, 8, (1, 4, 8, 8)), (32, 16, (1, 5, 8, 4))])
def test_halo_kernel_layers_vs_torch_cpu(cin, cout, shape, dt, tol):
    """3x3x3 stride-1 layers that take the LDS-halo kernel (Cg*size in {32,64} bytes, Cn <= 32), incl. ragged tiles."""
    torch.manual_seed(cin * 100 + cout)
    mod = torch.nn.Conv3d(cin, cout, 3, 1, 1)
    x = torch.randn(shape[0], cin, *shape[1:])
    xr = x.clone().requires_grad_(True)
    y = mod(xr)
    gy = torch.randn_like(y)
    y.backward(gy)
    yy, gx, gw, gb = run_conv_module(mod, x, gy, dt, DEV)
    assert rel_err(yy, y.detach()) < tol, "forward"
    assert rel_err(gx, xr.grad) < tol, "input gradient"
    assert rel_err(gw, mod.weight.grad) < tol, "weight gradient"
    assert rel_err(gb, mod.bias.grad) < tol, "bias gradient"


@pytest.mark.parametrize("kind,cin,cout,shape", [("convT", 64, 10, (1, 8, 8, 8)), ("convT", 32, 8, (2, 5, 9, 12)),
                                                 ("convT", 64, 16, (1, 4, 8, 4)), ("conv_s2", 8, 32, (1, 10, 16, 8)),
                                                 ("conv_s2", 16, 64, (2, 8, 8, 16))])
def test_up_halo_kernel_vs_torch_cpu(kind, cin, cout, shape):
    """8-class stride-2 passes that take conv_up_halo (bf16, gathered channels*2 in {64,128} bytes, <= 16 columns):
    ConvTranspose3d forward, and the input gradient of a stride-2 Conv3d; ragged tiles included."""
    torch.manual_seed(cin + cout)
    if kind == "convT":
        mod = torch.nn.ConvTranspose3d(cin, cout, 3, 2, 1, output_padding=1)
    else:
        mod = torch.nn.Conv3d(cin, cout, 3, 2, 1)
    x = torch.randn(shape[0], cin, *shape[1:])
    xr = x.clone().requires_grad_(True)
    y = mod(xr)
    gy = torch.randn_like(y)
    y.backward(gy)
    yy, gx, gw, gb = run_conv_module(mod, x, gy, BF16, DEV)
    assert rel_err(yy, y.detach()) < 2.5e-2, "forward"
    assert rel_err(gx, xr.grad) < 2.5e-2, "input gradient"
    assert rel_err(gw, mod.weight.grad) < 2.5e-2, "weight gradient"
    assert rel_err(gb, mod.bias.grad) < 2.5e-2, "bias gradient"


@pytest.mark.parametrize("kind,cin,cout,shape", [("conv", 64, 64, (2, 16, 16, 12)), ("conv", 64, 64, (1, 9, 20, 13)),
                                                 ("conv", 64, 64, (1, 16, 16, 8)), ("convT", 128, 32, (1, 8, 16, 16)),
                                                 ("convT", 128, 32, (2, 6, 20, 12)), ("conv_s2", 32, 128, (1, 32, 32, 16)),
                                                 ("conv_s2", 32, 128, (1, 18, 40, 24))])
def test_streamed_weight_halo_kernel_vs_torch_cpu(kind, cin, cout, shape):
    _check_conv_module(kind, cin, cout, shape)


@pytest.mark.parametrize("kind,cin,cout,shape", [("conv", 64, 64, (1, 1, 50, 48)), ("conv", 64, 64, (1, 300, 3, 3)),
                                                 ("conv", 64, 64, (1, 5, 5, 100)), ("convT", 128, 32, (1, 1, 40, 52)),
                                                 ("conv_s2", 16, 64, (1, 10, 34, 50)), ("conv_s2", 16, 64, (3, 2, 66, 66)),
                                                 ("convT", 64, 16, (1, 17, 11, 13))])
def test_halo_kernels_on_degenerate_extents(kind, cin, cout, shape):
    """extents of 1-5 voxels, long thin volumes, odd sizes: every tile of the streamed-weight / stride-2 halo kernels is ragged"""
    _check_conv_module(kind, cin, cout, shape)


@pytest.mark.parametrize("kind,cin,cout,shape", [("conv", 256, 256, (1, 8, 8, 6)), ("conv", 128, 256, (2, 7, 9, 5)),
                                                 ("conv", 96, 160, (1, 6, 10, 7)), ("conv_s2", 64, 256, (1, 12, 14, 10)),
                                                 ("convT", 384, 64, (1, 5, 6, 4)), ("conv", 256, 128, (1, 9, 7, 6)),
                                                 ("conv", 128, 128, (2, 8, 8, 6)), ("conv_s2", 32, 128, (1, 20, 12, 10)),
                                                 ("conv", 64, 96, (1, 7, 9, 11))])
def test_ring_pipelined_tile_vs_torch_cpu(kind, cin, cout, shape):
    """passes with > 128 output columns and a multiple of 32 gathered channels (bf16) take conv_igemm_ring_kernel: forward of
    Conv3d n->256 / 96->160 (column tile 5/8 full), input gradient of 256->128 (Cn = 256) and of the stride-2 64->256
    (8 parity classes, Cn = 64: stays on another kernel), input gradient of ConvTranspose3d 384->64 (stride-2 gather,
    Cn = 384 = 1.5 column tiles); 65..128 columns take the 192 x 128 tile of the same kernel (128->128, 256->128, the stride-2
    32->128 and a 3/4-full 64->96).  Every row tile is ragged (rows < 192 or not a multiple), padding taps on every face."""
    _check_conv_module(kind, cin, cout, shape)


@pytest.mark.parametrize("kind,cin,cout,shape", [("convT", 384, 64, (1, 16, 16, 8)), ("convT", 384, 64, (2, 10, 18, 12)),
                                                 ("convT", 256, 64, (1, 6, 20, 18)), ("conv_s2", 64, 256, (1, 36, 40, 12)),
                                                 ("conv_s2", 64, 384, (2, 32, 32, 8)), ("convT", 128, 64, (1, 3, 30, 26))])
def test_many_channel_8_class_kernel_vs_torch_cpu(kind, cin, cout, shape):
    """8-class stride-2 passes with >= 128 gathered channels and 64 columns take conv_up8_kernel (all classes' accumulators live, K
    walked in 32-channel chunks, delta-major taps): ConvTranspose3d 384 / 256 / 128 -> 64 forward and the input gradient of a stride-2
    Conv3d 64 -> 256 / 384 (Cg = 256 / 384).  >= 2048 coarse voxels, ragged 2 x 8 x 8 tiles on every axis, both tile-axis mappings,
    1-2 samples."""
    _check_conv_module(kind, cin, cout, shape)


@pytest.mark.parametrize("kind,cin,cout,shape", [("conv", 64, 64, (1, 16, 16, 16)), ("conv_s2", 32, 128, (2, 32, 32, 16)),
                                                 ("convT", 64, 16, (2, 16, 16, 16)), ("conv", 128, 256, (1, 16, 16, 16))])
def test_weight_gradient_with_xcd_grouped_slabs_vs_torch_cpu(kind, cin, cout, shape):
    """shapes whose weight gradient runs N * splits = 8 or 16 slabs: the generic kernel then takes its 1-D grid in which the
    K / column blocks of one slab are workgroups L, L+8, L+16 ... (one XCD per slab); 14 / 7 / 4 / 27 K blocks, 1-2 column blocks"""
    _check_conv_module(kind, cin, cout, shape)


def _check_conv_module(kind, cin, cout, shape):
    """passes that take conv_halo_sw (bf16): 64->64 k3 s1 forward + input gradient (one class), ConvTranspose3d 128->32
    forward and the input gradient of a stride-2 conv 32->128 (8 parity classes); both tile-axis mappings, ragged tiles."""
    torch.manual_seed(cin + cout + shape[3])
    if kind == "convT":
        mod = torch.nn.ConvTranspose3d(cin, cout, 3, 2, 1, output_padding=1)
    else:
        mod = torch.nn.Conv3d(cin, cout, 3, 2 if kind == "conv_s2" else 1, 1)
    x = torch.randn(shape[0], cin, *shape[1:])
    xr = x.clone().requires_grad_(True)
    y = mod(xr)
    gy = torch.randn_like(y)
    y.backward(gy)
    yy, gx, gw, gb = run_conv_module(mod, x, gy, BF16, DEV)
    assert rel_err(yy, y.detach()) < 2.5e-2, "forward"
    assert rel_err(gx, xr.grad) < 2.5e-2, "input gradient"
    assert rel_err(gw, mod.weight.grad) < 2.5e-2, "weight gradient"
    assert rel_err(gb, mod.bias.grad) < 2.5e-2, "bias gradient"


@pytest.mark.parametrize("kind,cin,cout,shape", [("conv_s2", 32, 128, (1, 32, 32, 24)), ("conv_s2", 32, 64, (2, 18, 40, 34)),
                                                 ("conv_s2", 16, 64, (1, 32, 32, 16)), ("conv_s2", 16, 48, (1, 22, 36, 50)),
                                                 ("convT", 64, 16, (1, 16, 16, 8)), ("convT", 128, 32, (1, 16, 16, 12))])
def test_stride2_halo_kernel_vs_torch_cpu(kind, cin, cout, shape):
    """stride-2 passes that take conv_down_halo (bf16, 16 / 32 gathered channels, >= 48 columns): Conv3d k3 s2 forward and the
    input gradient of a ConvTranspose3d (a stride-2 conv over dOut); odd extents, both tile-axis mappings, two column blocks."""
    torch.manual_seed(cin * 3 + cout + shape[2])
    if kind == "convT":
        mod = torch.nn.ConvTranspose3d(cin, cout, 3, 2, 1, output_padding=1)
    else:
        mod = torch.nn.Conv3d(cin, cout, 3, 2, 1)
    x = torch.randn(shape[0], cin, *shape[1:])
    xr = x.clone().requires_grad_(True)
    y = mod(xr)
    gy = torch.randn_like(y)
    y.backward(gy)
    yy, gx, gw, gb = run_conv_module(mod, x, gy, BF16, DEV)
    assert rel_err(yy, y.detach()) < 2.5e-2, "forward"
    assert rel_err(gx, xr.grad) < 2.5e-2, "input gradient"
    assert rel_err(gw, mod.weight.grad) < 2.5e-2, "weight gradient"
    assert rel_err(gb, mod.bias.grad) < 2.5e-2, "bias gradient"


@pytest.mark.parametrize("cout,shape", [(16, (1, 16, 16, 8)), (64, (2, 8, 16, 16)), (32, (1, 10, 12, 8)), (48, (1, 8, 8, 24))])
def test_stem_kernels_vs_torch_cpu(cout, shape):
    """single-channel 3x3x3 stride-2 conv (conv_stem.hip: forward with LDS input patch, weight gradient with 4 slabs per
    workgroup), bf16, ragged tiles included."""
    torch.manual_seed(cout)
    mod = torch.nn.Conv3d(1, cout, 3, 2, 1)
    x = torch.randn(shape[0], 1, *shape[1:])
    y = mod(x)
    gy = torch.randn_like(y)
    y.backward(gy)
    yy, gx, gw, gb = run_conv_module(mod, x, gy, BF16, DEV)
    assert gx is None
    assert rel_err(yy, y.detach()) < 2.5e-2, "forward"
    assert rel_err(gw, mod.weight.grad) < 2.5e-2, "weight gradient"
    assert rel_err(gb, mod.bias.grad) < 2.5e-2, "bias gradient"


@pytest.mark.parametrize("dt,tol", [(F32, 1e-5), (BF16, 2e-2)])
def test_instnorm_prelu_fwd_bwd(golden, dt, tol):
    """conv-epilogue statistics -> finalize -> apply, and the 3-kernel backward, against InstanceNorm3d+PReLU."""
    from capstone_amd.engine import GemmLayer, rup
    from capstone_amd.plan import _NormAct
    g = golden("ops_torch.npz")
    x, gy = torch.from_numpy(g["in_prelu_x"]), torch.from_numpy(g["in_prelu_gy"])
    C = x.shape[1]
    # identity 1x1x1 conv so that the statistics come out of the conv epilogue exactly as in the network
    conv = torch.nn.Conv3d(C, C, 1)
    alpha = torch.nn.Parameter(torch.from_numpy(g["in_prelu_1.weight"]).clone())
    with torch.no_grad():
        conv.weight.copy_(torch.eye(C).reshape(C, C, 1, 1, 1))
        conv.bias.zero_()
    plan = MiniPlan([conv.weight, conv.bias, alpha], DEV, dt, 3)
    layer = GemmLayer(plan, "id", False, 1, 1, C, [(conv.weight, conv.bias, C)], C)
    plan.packer.finalize()
    xa = to_cl(x, dt, DEV)
    y, stats = layer.emit_fwd(xa, want_stats=True)
    na = _NormAct(plan, alpha)
    out = na.emit_fwd(y, stats, 0, None, None)
    plan.run()
    xin = xa.valid().float().cpu()                       # what the kernel actually normalised (bf16-rounded input)
    ref = torch.nn.Sequential(torch.nn.InstanceNorm3d(C), torch.nn.PReLU())
    with torch.no_grad():
        ref[1].weight.copy_(alpha.detach().cpu())
    xr = xin.clone().requires_grad_(True)
    yr = ref(xr)
    yr.backward(gy)
    assert rel_err(from_cl(out), yr.detach()) < tol
    ga = to_cl(gy, dt, DEV)
    dy = na.emit_bwd(ga)
    plan.run()
    torch.cuda.synchronize()
    assert rel_err(from_cl(dy), xr.grad) < max(tol, 3e-5)
    da = plan.store.grad_view(alpha).cpu()
    assert abs(float(da) - float(ref[1].weight.grad)) < max(tol, 1e-5) * max(1.0, abs(float(ref[1].weight.grad)))


def test_reference_leaf_fixtures_on_gpu(golden):
    """squash / softmax-argmax (incl. engineered ties) / Dice metric: bit-exact vs the reference's own outputs."""
    from capstone_amd.training.utils import _squash_predictions
    from capstone_amd.volumetric.metrics import DiceMetricWrapper3D
    from capstone_amd.volumetric.utils import _squash_masks_3D
    leaf = golden("ref_leaf.npz")
    got = _squash_masks_3D(torch.from_numpy(leaf["squash_masks_in"]).to(DEV), 10, DEV)
    assert got.dtype == torch.int64
    np.testing.assert_array_equal(got.cpu().numpy(), leaf["squash_masks_out"])
    pred = _squash_predictions(torch.from_numpy(leaf["squash_pred_in"]).to(DEV))
    np.testing.assert_array_equal(pred.cpu().numpy(), leaf["squash_pred_out"])
    m, pc = DiceMetricWrapper3D()(torch.from_numpy(leaf["dice_pred"]).to(DEV), torch.from_numpy(leaf["dice_target"]).to(DEV))
    np.testing.assert_allclose(pc.cpu().numpy(), leaf["dice_per_class"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(m.cpu().numpy(), leaf["dice_mean"], rtol=0, atol=1e-7)


def test_losses_vs_reference_and_oracle(golden):
    from capstone_amd.models.losses import MultipleLossWrapper
    from oracle import losses as OL
    leaf = golden("ref_leaf.npz")
    lg, tg = torch.from_numpy(leaf["ce_logits"]), torch.from_numpy(leaf["ce_target"])
    for name, vkey, gkey in (("CrossEntropy", "ce_value", "ce_grad"), ("WeightedCrossEntropy", "wce_value", "wce_grad")):
        x = lg.to(DEV).requires_grad_(True)
        v = MultipleLossWrapper([name])(input=x, target=tg.to(DEV))[name]
        v.backward()
        np.testing.assert_allclose(v.item(), leaf[vkey], rtol=2e-6)
        np.testing.assert_allclose(x.grad.cpu().numpy(), leaf[gkey], rtol=1e-4, atol=1e-9)
    names = ["Dice", "Focal", "GeneralizedDice"]
    tg2 = torch.from_numpy(leaf["gdl_target"])
    x = lg.to(DEV).requires_grad_(True)
    vals = MultipleLossWrapper(names)(input=x, target=tg2.to(DEV))
    torch.stack(list(vals.values())).sum().backward()
    np.testing.assert_allclose(vals["GeneralizedDice"].item(), leaf["gdl_mean"], rtol=1e-5)   # reference-local GDL
    xr = lg.clone().requires_grad_(True)
    rv = OL.MultipleLoss(names)(xr, tg2)
    torch.stack(list(rv.values())).sum().backward()
    for n in names:
        np.testing.assert_allclose(vals[n].item(), rv[n].item(), rtol=1e-5, err_msg=n)
    np.testing.assert_allclose(x.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-3, atol=1e-8)


def _load_tiny(golden, tag, precision):
    from capstone_amd.volumetric.base_trainer import BaseUNet3D
    g = golden("unet_tiny.npz")
    m = BaseUNet3D(filters=[int(v) for v in g[f"{tag}_filters"]], loss_fx=[str(s) for s in g[f"{tag}_losses"]], precision=precision)
    m.load_state_dict({k[len(tag) + 3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"{tag}_w:")})
    m.to(DEV)
    batch = tuple(torch.from_numpy(g[f"{tag}_{n}"]).to(DEV) for n in ("images", "masks", "indicator"))
    return g, m, batch


def _assert_adam_moved_like_fixture(w_got, w0, w1_ref, g_ref, key, lr=1e-3):
    """Adam's first step moves a weight by lr * g / (|g| + eps) ~ -lr * sign(g).  For every element whose fixture gradient is
    above the noise floor: the step taken has the fixture's SIGN and its size (|w1 - w0| ~ lr), i.e. a step that never ran, ran
    twice, or ran on the wrong gradient fails.  (Biases feeding an InstanceNorm hold rounding noise only: excluded.)  Returns the
    number of elements checked."""
    if key.endswith(".bias") and "residual" not in key:
        return 0
    solid = np.abs(g_ref) > 1e-3 * max(float(np.abs(g_ref).max()), 1e-6)
    if not solid.any():
        return 0
    step_got, step_ref = (w_got - w0)[solid], (w1_ref - w0)[solid]
    assert np.all(np.abs(step_ref) > 0.5 * lr), key                # the fixture itself moved these by ~lr
    np.testing.assert_array_equal(np.sign(step_got), np.sign(step_ref), err_msg=key + " (direction of the Adam step)")
    np.testing.assert_allclose(np.abs(step_got), np.abs(step_ref), rtol=0, atol=0.05 * lr, err_msg=key + " (size of the Adam step)")
    return int(solid.sum())


@pytest.mark.parametrize("tag", ["a", "b"])
def test_tiny_step_fp32_matches_oracle_fixture(golden, tag):
    g, m, batch = _load_tiny(golden, tag, "fp32")
    loss = m.fit_step(batch)
    eng = m.unet.engine()
    logits = eng.logits_view().cpu().numpy()
    assert np.abs(logits - g[f"{tag}_logits"]).max() < 1e-3                       # north_star: logits within 1e-3 fp32
    assert np.abs(logits - g[f"{tag}_logits"]).max() < 5e-5
    np.testing.assert_allclose(loss.item(), g[f"{tag}_loss"], rtol=1e-5)
    np.testing.assert_allclose(m.logged["Mean Dice Score (train)"].item(), g[f"{tag}_dice_mean"], atol=2e-3)
    np.testing.assert_allclose(m.logged["Dice per class (train)"].cpu().numpy(), g[f"{tag}_dice_per_class"], atol=2e-3)
    checked = 0
    for k, p in m.named_parameters():
        ref = g[f"{tag}_g:{k}"]
        scale = max(1.0, float(np.abs(ref).max()))
        atol = (5e-3 if (k.endswith(".bias") and "residual" not in k) else 5e-5) * scale
        np.testing.assert_allclose(eng.store.grad_view(p).cpu().numpy(), ref, rtol=3e-3, atol=atol, err_msg=k)
        checked += _assert_adam_moved_like_fixture(p.detach().cpu().numpy(), g[f"{tag}_w:{k}"], g[f"{tag}_w1:{k}"], ref, k)
    assert checked > 0.5 * sum(p.numel() for p in m.parameters()), "the post-Adam check must cover most of the weights"


def test_masks_bit_exact_where_margin(golden):
    """argmax masks: identical to the CPU oracle on every voxel whose top-2 logit margin exceeds the logit error."""
    from capstone_amd.training.utils import _squash_predictions
    from oracle.metrics import squash_predictions
    g, m, batch = _load_tiny(golden, "b", "fp32")
    logits = m(batch[0])
    ref = torch.from_numpy(g["b_logits"])
    err = float((logits.cpu() - ref).abs().max())
    top2 = ref.topk(2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > 4 * max(err, 1e-6)
    got, exp = _squash_predictions(logits).cpu(), squash_predictions(ref)
    assert safe.float().mean() > 0.99
    assert torch.equal(got[safe], exp[safe])
    # and on IDENTICAL logits the mask kernel is bit-exact everywhere
    assert torch.equal(_squash_predictions(ref.to(DEV)).cpu(), exp)


def test_autograd_drop_in_path_equals_native_step(golden):
    """training_step -> loss.backward() -> torch.optim.Adam.step()  ==  fit_step (native order)."""
    g, m1, batch = _load_tiny(golden, "b", "fp32")
    _, m2, _ = _load_tiny(golden, "b", "fp32")
    opt = m1.configure_optimizers()
    opt.zero_grad()
    loss = m1.training_step(batch, 0)
    loss.backward()
    opt.step()
    loss2 = m2.fit_step(batch)
    np.testing.assert_allclose(loss.item(), loss2.item(), rtol=1e-6)
    checked = 0
    for (k, p), q in zip(m1.named_parameters(), m2.parameters()):
        # both paths replay the same recorded conv / norm programs; only the loss gradient is produced by different passes (stats +
        # gradient vs the fused CE pass), so gradients agree to rounding and, wherever the gradient is not itself rounding noise,
        # torch's Adam and ctseg_adam_step land within a small fraction of one step (lr = 1e-3) of each other
        w0, wa, wn = g[f"b_w:{k}"], p.detach().cpu().numpy(), q.detach().cpu().numpy()
        gref = g[f"b_g:{k}"]
        real = np.abs(gref) > 1e-4 * max(float(np.abs(gref).max()), 1e-6)
        if k.endswith(".bias") and "residual" not in k:
            real &= False
        np.testing.assert_allclose(wa[real], wn[real], rtol=0, atol=2e-5, err_msg=k)
        np.testing.assert_allclose(wa, wn, rtol=0, atol=2.1e-3, err_msg=k)
        checked += _assert_adam_moved_like_fixture(wn, w0, g[f"b_w1:{k}"], g[f"b_g:{k}"], k)
        if real.any():
            assert np.abs(wa - w0)[real].max() > 0.5e-3, k          # and a step was taken at all
    assert checked > 0.5 * sum(p.numel() for p in m1.parameters())
    # second forward after the torch optimizer changed the weights in place: packed operands must refresh
    l1b = m1.training_step(batch, 1).item()
    l2b = m2.fit_step(batch).item()
    assert abs(l1b - l2b) < 5e-3 * max(1.0, abs(l2b))
    assert l2b < loss2.item()


def test_bf16_step_close_to_oracle(golden):
    """bf16 storage + MFMA: same step, bounded drift (logits rel. L-inf < 6e-2, loss within 2 %, Dice +-0.002 is an fp32 claim;
    here the bf16 Dice must stay within 0.02 of the fixture on a random-init net whose logits are near-ties)."""
    g = golden("unet_tiny.npz")
    from capstone_amd.volumetric.base_trainer import BaseUNet3D
    import oracle.trainer as OT
    torch.manual_seed(7)
    om = OT.OracleUNet3D(filters=(8, 16, 32, 64), loss_fx=("CrossEntropy",))
    m = BaseUNet3D(filters=[8, 16, 32, 64], loss_fx=["CrossEntropy"], precision="bf16")
    m.load_state_dict(om.state_dict())
    m.to(DEV)
    batch = tuple(torch.from_numpy(g[f"b_{n}"]) for n in ("images", "masks", "indicator"))
    ologits = om(batch[0])
    oloss = om.training_step(batch)
    oloss.backward()
    loss = m.fit_step(tuple(t.to(DEV) for t in batch))
    eng = m.unet.engine()
    assert rel_err(eng.logits_view().cpu(), ologits.detach()) < 6e-2
    assert abs(loss.item() - oloss.item()) < 2e-2 * oloss.item()
    cos = []
    for (k, p), q in zip(om.named_parameters(), m.parameters()):
        a, b = eng.store.grad_view(q).cpu().flatten().double(), p.grad.flatten().double()
        if b.norm() > 1e-3:
            cos.append(float(torch.dot(a, b) / (a.norm() * b.norm())))
    assert min(cos) > 0.98, min(cos)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_config_c_step_vs_oracle(precision):
    """BASELINE.json configs[1]: the real network (32,64,128,256) on 1x1x128x128x32, full training step against the CPU oracle.
    fp32: logits within 1e-3 (north_star), loss 1e-4, mean Dice +-0.002, masks equal wherever the top-2 margin exceeds the
    logit error.  bf16: loss within 2 %, gradient direction (cosine) > 0.97 per tensor."""
    from bench import synthetic_batch
    from capstone_amd.volumetric.base_trainer import BaseUNet3D
    from oracle.metrics import squash_predictions
    import oracle.trainer as OT
    torch.manual_seed(12342)
    om = OT.OracleUNet3D(filters=(32, 64, 128, 256), loss_fx=("CrossEntropy",))
    m = BaseUNet3D(filters=[32, 64, 128, 256], loss_fx=["CrossEntropy"], precision=precision)
    m.load_state_dict(om.state_dict())
    m.to(DEV)
    batch = synthetic_batch(1, 128, 128, 32, "cpu", 12342)
    _, _, _, ologits, oloss = om.shared_step(batch, True)
    oloss.backward()
    loss = m.fit_step(tuple(t.to(DEV) for t in batch))
    eng = m.unet.engine()
    logits = eng.logits_view().cpu()
    err = float((logits - ologits.detach()).abs().max())
    odice = float(om.logged["Mean Dice Score (train)"])
    dice = float(m.logged["Mean Dice Score (train)"])
    if precision == "fp32":
        assert err < 1e-3, err
        assert abs(loss.item() - oloss.item()) < 1e-4 * abs(oloss.item())
        assert abs(dice - odice) <= 0.002
        top2 = ologits.detach().topk(2, dim=1).values
        safe = (top2[:, 0] - top2[:, 1]) > 4 * max(err, 1e-6)
        pred = torch.softmax(logits, 1).argmax(1)
        assert torch.equal(pred[safe], squash_predictions(ologits.detach())[safe])
    else:
        assert abs(loss.item() - oloss.item()) < 2e-2 * abs(oloss.item())
        assert abs(dice - odice) <= 0.02
    cos = []
    for (k, p), q in zip(om.named_parameters(), m.parameters()):
        a, b = eng.store.grad_view(q).cpu().flatten().double(), p.grad.flatten().double()
        if b.norm() > 1e-4:
            cos.append((float(torch.dot(a, b) / (a.norm() * b.norm())), k))
    assert min(cos)[0] > (0.9999 if precision == "fp32" else 0.97), min(cos)


@pytest.mark.parametrize("nres", [0, 2])
def test_2d_path_and_validation_step(nres):
    """BASELINE.json configs[0] (2-D U-Net on a CT slice; 128x128 here to keep the CPU oracle fast), fp32: the 2-D model is the
    3-D engine with Z = 1.  training_step through autograd + validation_step (no-grad path) vs the oracle."""
    from capstone_amd.training.base_trainer import BaseUNet2D
    from oracle import losses as OL, metrics as OM
    from oracle.monai_unet import UNet as OracleUNet
    torch.manual_seed(5)
    filters = [8, 16, 32, 64, 128]
    ref = OracleUNet(2, 1, 10, filters, (2, 2, 2, 2), num_res_units=nres)
    m = BaseUNet2D(filters=list(filters), use_res_units=nres > 0, loss_fx=["Focal", "Dice"], transform_degree=0)
    m.unet.load_state_dict(ref.state_dict())
    m.to(DEV)
    g = torch.Generator().manual_seed(6)
    images = torch.randn(2, 1, 128, 128, generator=g)
    masks = torch.zeros(2, 9, 128, 128, dtype=torch.uint8)
    for c in range(9):
        masks[:, c, 10 * c + 5:10 * c + 20, 30:90] = 1
    ind = torch.ones(2, 9)
    labels = OM.squash_masks(masks, 10)
    y_ref = ref(images)
    rv = OL.MultipleLoss(["Dice", "Focal"])(y_ref, labels, ind)
    total_ref = torch.stack(list(rv.values())).sum()
    total_ref.backward()
    batch = (images.to(DEV), masks.to(DEV), ind.to(DEV))
    loss = m.training_step(batch)
    loss.backward()
    np.testing.assert_allclose(loss.item(), total_ref.item(), rtol=1e-4)
    odice, _ = OM.DiceMetric()(OM.squash_predictions(y_ref.detach()), labels)
    assert abs(m.logged["Mean Dice Score (train)"].item() - odice.item()) <= 0.002
    for (k, p), q in zip(ref.named_parameters(), m.unet.parameters()):
        a, b = q.grad.cpu().flatten().double(), p.grad.flatten().double()
        if b.norm() > 1e-5:
            assert float(torch.dot(a, b) / (a.norm() * b.norm())) > 0.9999, k
    with torch.no_grad():
        m.validation_step(batch)
    np.testing.assert_allclose(m.logged["Dice Loss (val)"].item(), rv["Dice"].item(), rtol=1e-4)
    assert abs(m.logged["Mean Dice Score (val)"].item() - odice.item()) <= 0.002
    opt = m.configure_optimizers()
    assert opt["monitor"] == "Mean Dice Score (val)" and isinstance(opt["lr_scheduler"], torch.optim.lr_scheduler.ReduceLROnPlateau)


def test_3d_validation_step_no_grad(golden):
    g, m, batch = _load_tiny(golden, "b", "fp32")
    with torch.no_grad():
        m.validation_step(batch, 0)
    np.testing.assert_allclose(m.logged["CrossEntropy Loss (val)"].item() + m.logged["Dice Loss (val)"].item(), g["b_loss"], rtol=1e-4)
    np.testing.assert_allclose(m.logged["Mean Dice Score (val)"].item(), g["b_dice_mean"], atol=2e-3)


def test_adam_matches_torch():
    torch.manual_seed(0)
    n = 10007
    p0, g0 = torch.randn(n), torch.randn(n)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=1e-3)
    from capstone_amd.engine import ParamStore
    q = torch.nn.Parameter(p0.clone())
    st = ParamStore([q], torch.device(DEV))
    for step in range(3):
        ref.grad = g0 * (step + 1)
        opt.step()
        st.flat_g[:n].copy_((g0 * (step + 1)).to(DEV))
        st.adam_step(1e-3)
    np.testing.assert_allclose(q.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-6, atol=1e-7)


def test_full_size_properties_bf16():
    """BASELINE.json's metric config (2 x 512 x 512 x 48, bf16): size-independent properties.
    counts partition the volume, the step is deterministic (bit-identical rerun), loss is finite and decreases."""
    from capstone_amd.volumetric.base_trainer import BaseUNet3D
    torch.manual_seed(12342)
    m = BaseUNet3D(filters=[32, 64, 128, 256], loss_fx=["CrossEntropy"], precision="bf16").to(DEV)
    g = torch.Generator(device=DEV).manual_seed(12342)
    B, H, W, D = 2, 512, 512, 48
    images = torch.randn(B, 1, H, W, D, device=DEV, generator=g)
    masks = torch.zeros(B, 9, H, W, D, dtype=torch.uint8, device=DEV)
    for c in range(9):
        masks[:, c, 40 * c + 20:40 * c + 50, 100:180, 8:30] = 1
    ind = torch.ones(B, 9, device=DEV)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    l0 = m.fit_step((images, masks, ind)).item()
    le = m.unet.engine().last_plan._ctseg_loss
    cnt = le.cnt.cpu()
    S = H * W * D
    assert torch.equal(cnt[:, 1].sum(1), torch.full((B,), S)) and torch.equal(cnt[:, 2].sum(1), torch.full((B,), S))
    assert torch.equal(cnt[:, 2, 1:], torch.full((B, 9), 30 * 80 * 22))
    assert (cnt[:, 0] <= torch.minimum(cnt[:, 1], cnt[:, 2])).all()
    g1 = m.unet.engine().store.flat_g.clone()
    l1 = m.fit_step((images, masks, ind)).item()
    assert np.isfinite(l0) and np.isfinite(l1) and l1 < l0
    # determinism: reload the initial weights, reset Adam, rerun -> identical gradient bits
    m.load_state_dict(sd)
    st = m.unet.engine().store
    st.adam_m.zero_(); st.adam_v.zero_(); st.step = 0
    l0b = m.fit_step((images, masks, ind)).item()
    assert l0b == l0 and torch.equal(st.flat_g, g1)


def test_side_stream_backward_is_bit_identical(monkeypatch):
    """weight gradients on the second HIP stream (default) vs everything on one stream: same bits in the flat gradient buffer
    and the same updated weights (no float atomics anywhere, so stream interleaving cannot change a sum order)."""
    from capstone_amd.volumetric.base_trainer import BaseUNet3D
    g = torch.Generator().manual_seed(11)
    images = torch.randn(2, 1, 32, 32, 16, generator=g).to(DEV)
    masks = (torch.rand(2, 9, 32, 32, 16, generator=g) < 0.1).to(torch.uint8).to(DEV)
    ind = torch.ones(2, 9, dtype=torch.float64).to(DEV)
    out = []
    for side in ("1", "0"):
        monkeypatch.setenv("CTSEG_SIDE_STREAM", side)
        torch.manual_seed(3)
        m = BaseUNet3D(filters=[16, 32, 64], loss_fx=["CrossEntropy"], precision="bf16").to(DEV)
        losses = [float(m.fit_step((images, masks, ind))) for _ in range(3)]
        st = m.unet.engine().store
        torch.cuda.synchronize()
        out.append((losses, st.flat_g.clone(), st.flat_p.clone()))
    assert out[0][0] == out[1][0]
    assert torch.equal(out[0][1], out[1][1]) and torch.equal(out[0][2], out[1][2])


def test_rccl_gradient_exchange_on_one_rank_matches_plain_step(monkeypatch):
    """backend "nccl" (= RCCL) with a one-rank group: broadcast, the chunked async all-reduce fired from the
    backward hooks (joined with the weight-gradient side stream), the stream hand-back in finish().  The sums
    are the identity on one rank, so losses / gradients / weights must equal the plain step bit for bit."""
    import socket
    import torch.distributed as dist
    from capstone_amd import distributed as cdist
    from capstone_amd.volumetric.base_trainer import BaseUNet3D
    g = torch.Generator().manual_seed(12)
    images = torch.randn(2, 1, 32, 32, 16, generator=g).to(DEV)
    masks = (torch.rand(2, 9, 32, 32, 16, generator=g) < 0.1).to(torch.uint8).to(DEV)
    ind = torch.ones(2, 9, dtype=torch.float64).to(DEV)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", str(port))
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    try:
        out = []
        # exchange: off / the chunked all-reduce / the direct exchange (all-to-all of shards -> rank-ordered sum -> all-gather on a
        # communication stream of its own, CTSEG_DDP_ALGO=direct): all three are the identity on one rank
        for exchange in (None, "allreduce", "direct"):
            if exchange:
                monkeypatch.setenv("CTSEG_DDP_ALGO", exchange)
            torch.manual_seed(4)
            m = BaseUNet3D(filters=[16, 32, 64], loss_fx=["CrossEntropy"], precision="bf16").to(DEV)
            losses = [float(m.fit_step((images, masks, ind)))]
            if exchange:
                red = cdist.attach(m, always=True)
                assert red.active and red.points and red.algo == exchange, "the readiness split must exist so a chunk goes out mid-backward"
            losses += [float(m.fit_step((images, masks, ind))) for _ in range(3)]
            st = m.unet.engine().store
            torch.cuda.synchronize()
            out.append((losses, st.flat_g.clone(), st.flat_p.clone()))
        for k in (1, 2):
            assert out[0][0] == out[k][0]
            assert torch.equal(out[0][1], out[k][1]) and torch.equal(out[0][2], out[k][2])
    finally:
        dist.destroy_process_group()


def test_narrow_rows_change_no_forward_bit_and_no_gradient_beyond_sum_order(monkeypatch):
    """bf16 tensors either side of the 10-class logits convolution are laid out 12 wide (24-byte rows) when every pass that
    touches them can move such rows.  The layout changes no arithmetic of the forward pass (logits bit-identical to the
    16-wide plan); the backward differs only in the order of the fp32 InstanceNorm-backward partial sums."""
    from capstone_amd.volumetric.base_trainer import BaseUNet3D
    # both layouts on the SAME kernels: 16-byte-chunked rows would otherwise take the x-column halo kernel (conv_halo_x.hip), whose
    # K order differs from the register-staged kernel the 12-wide rows run on (equal up to fp32 summation order, tested below)
    monkeypatch.setenv("CTSEG_NO_HALO_X", "1")
    g = torch.Generator().manual_seed(21)
    images = torch.randn(2, 1, 32, 48, 16, generator=g).to(DEV)
    masks = (torch.rand(2, 9, 32, 48, 16, generator=g) < 0.1).to(torch.uint8).to(DEV)
    ind = torch.ones(2, 9, dtype=torch.float64).to(DEV)
    out = {}
    for narrow in ("1", "0"):
        monkeypatch.setenv("CTSEG_NARROW_ROWS", narrow)
        torch.manual_seed(5)
        m = BaseUNet3D(filters=[16, 32, 64], loss_fx=["CrossEntropy"], precision="bf16").to(DEV)
        loss = float(m.fit_step((images, masks, ind)))
        eng = m.unet.engine()
        plan = eng.last_plan
        assert plan.dlogits.ld == (12 if narrow == "1" else 16), "the 32x48x16 head is eligible for 12-wide rows"
        torch.cuda.synchronize()
        out[narrow] = (loss, eng.logits_view(plan).clone(), eng.store.flat_g.clone())
    assert out["1"][0] == out["0"][0]
    assert torch.equal(out["1"][1], out["0"][1])
    ga, gb = out["1"][2], out["0"][2]
    assert float((ga - gb).abs().max()) <= 2e-3 * float(gb.abs().max())


def test_x_column_halo_kernel_equals_register_staged_kernel_up_to_sum_order(monkeypatch):
    """conv_halo_x.hip (LDS-DMA staging, x-column fragment reuse, weights in registers) against conv_halo.hip on a whole training
    step in the 16-wide layout: same operands, same fp32 accumulators, only the order of the K steps differs."""
    from capstone_amd.volumetric.base_trainer import BaseUNet3D
    monkeypatch.setenv("CTSEG_NARROW_ROWS", "0")
    g = torch.Generator().manual_seed(23)
    images = torch.randn(2, 1, 32, 48, 16, generator=g).to(DEV)
    masks = (torch.rand(2, 9, 32, 48, 16, generator=g) < 0.1).to(torch.uint8).to(DEV)
    ind = torch.ones(2, 9, dtype=torch.float64).to(DEV)
    out = {}
    for off in ("0", "1"):
        if off == "1":
            monkeypatch.setenv("CTSEG_NO_HALO_X", "1")
        else:
            monkeypatch.delenv("CTSEG_NO_HALO_X", raising=False)
        torch.manual_seed(5)
        m = BaseUNet3D(filters=[32, 64, 128], loss_fx=["CrossEntropy"], precision="bf16").to(DEV)
        loss = float(m.fit_step((images, masks, ind)))
        eng = m.unet.engine()
        torch.cuda.synchronize()
        out[off] = (loss, eng.logits_view().clone(), eng.store.flat_g.clone())
    assert abs(out["0"][0] - out["1"][0]) <= 1e-4 * abs(out["1"][0])
    la, lb = out["0"][1], out["1"][1]
    assert float((la - lb).abs().max()) <= 2e-2 * float(lb.abs().max())      # bf16 re-rounding of activations downstream
    ga, gb = out["0"][2], out["1"][2]
    cos = float(torch.dot(ga.double(), gb.double()) / (ga.double().norm() * gb.double().norm()))
    assert cos > 0.9995, cos


def test_narrow_rows_fall_back_where_a_pass_cannot_move_them(monkeypatch):
    """depth 2 (< 4 voxels): the logits convolution does not take the LDS-halo kernel -> the plan is recorded 16 wide"""
    from capstone_amd.volumetric.base_trainer import BaseUNet3D
    monkeypatch.setenv("CTSEG_NARROW_ROWS", "1")
    g = torch.Generator().manual_seed(22)
    images = torch.randn(1, 1, 16, 16, 2, generator=g).to(DEV)
    masks = (torch.rand(1, 9, 16, 16, 2, generator=g) < 0.1).to(torch.uint8).to(DEV)
    ind = torch.ones(1, 9, dtype=torch.float64).to(DEV)
    torch.manual_seed(6)
    m = BaseUNet3D(filters=[16, 32], loss_fx=["CrossEntropy"], precision="bf16").to(DEV)
    loss = float(m.fit_step((images, masks, ind)))
    assert np.isfinite(loss) and m.unet.engine().last_plan.dlogits.ld == 16


def test_loss_dice_summary_kernel_matches_the_torch_expressions():
    """ctseg_loss_dice_summary (one launch) against SegLossEngine.loss_values + dice_metric (the torch expressions it replaces),
    with classes absent from some / all samples (NaN -> excluded from the batch mean, 0 when no sample holds the class)"""
    B, S, C = 3, 4096, 10
    le = segloss.SegLossEngine(torch.device(DEV), B, S, C)
    g = torch.Generator().manual_seed(31)
    le.red.copy_(torch.rand(B, le.R, generator=g, dtype=torch.float64) * 100 + 1)
    cnt = torch.randint(0, 500, (B, 3, C), generator=g, dtype=torch.int64)
    cnt[:, 2, 3] = 0            # class 3 absent everywhere
    cnt[1, 2, 5] = 0            # class 5 absent in one sample
    cnt[:, 0] = torch.minimum(cnt[:, 0], torch.minimum(cnt[:, 1], cnt[:, 2]))
    le.cnt.copy_(cnt)
    le.hist = cnt[:, 2].clone().to(DEV)
    want_loss = le.loss_values(["CrossEntropy"])["CrossEntropy"]
    want_mean, want_pc = le.dice_metric()
    loss, dm, dpc = le.ce_summary()
    torch.cuda.synchronize()
    assert abs(float(loss) - float(want_loss)) <= 1e-6 * abs(float(want_loss))
    assert torch.allclose(dpc.cpu(), want_pc.cpu(), rtol=1e-6, atol=1e-7) and float(dpc[2]) == 0.0
    assert abs(float(dm) - float(want_mean)) <= 1e-6


def test_many_channel_8_class_kernel_statistics_and_addend():
    """conv_up8_kernel's two extras, at op level: the InstanceNorm partial sums of the forward pass (one slot per workgroup and class
    group; their total = sum / sum of squares of the fp32 results over the voxels) and the addend of the input-gradient pass
    (dx = g + dgrad(dy), what the dense skip gradient uses).  Ragged 3 x 8 x 8 tiles, two samples."""
    from capstone_amd.engine import GemmLayer
    from helpers import MiniPlan, from_cl, rel_err, to_cl
    torch.manual_seed(5)
    mod = torch.nn.ConvTranspose3d(256, 64, 3, 2, 1, output_padding=1)
    x = torch.randn(2, 256, 10, 18, 12)
    with torch.no_grad():
        yref = mod(x)                      # (MiniPlan moves the parameters to the device)
    plan = MiniPlan([mod.weight, mod.bias], DEV, BF16, 3)
    xa = to_cl(x, BF16, DEV)
    layer = GemmLayer(plan, "t", True, 3, 2, 256, [(mod.weight, mod.bias, 64)], 256)
    plan.packer.finalize()
    y, stats = layer.emit_fwd(xa, want_stats=True)
    plan.run()
    torch.cuda.synchronize()
    assert rel_err(from_cl(y, False), yref) < 2.5e-2
    part = stats.partials.cpu()
    assert rel_err(part[:, :, 0, :64].sum(1), yref.sum(dim=(2, 3, 4))) < 2.5e-2
    assert rel_err(part[:, :, 1, :64].sum(1), (yref * yref).sum(dim=(2, 3, 4))) < 2.5e-2

    mod2 = torch.nn.Conv3d(64, 256, 3, 2, 1)
    x2 = torch.randn(2, 64, 20, 36, 12)
    xr = x2.clone().requires_grad_(True)
    y2 = mod2(xr)
    gy = torch.randn_like(y2)
    y2.backward(gy)
    addend = torch.randn_like(x2)
    plan2 = MiniPlan([mod2.weight, mod2.bias], DEV, BF16, 3)
    layer2 = GemmLayer(plan2, "t", False, 3, 2, 64, [(mod2.weight, mod2.bias, 256)], 64)
    plan2.packer.finalize()
    layer2.emit_fwd(to_cl(x2, BF16, DEV))         # (records the input geometry the backward passes use)
    ga, aa = to_cl(gy, BF16, DEV), to_cl(addend, BF16, DEV)
    gxa = layer2.emit_dgrad(ga, add=aa)
    plan2.run()
    torch.cuda.synchronize()
    ref = xr.grad + addend.bfloat16().float()
    assert rel_err(from_cl(gxa, False), ref) < 2.5e-2
