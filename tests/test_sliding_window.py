"""Sliding-window inference: product vs the MONAI-0.3 restatement in oracle/sliding_window.py.

CPU: window grid + importance maps (host logic) and the whole inferer through the numpy ABI emulator.
GPU (-m gpu): the same comparison through the HIP kernels, fp32 and bf16, and the SlidingWindowInferer surface."""
import numpy as np
import pytest
import torch

from capstone_amd import _native as nat
from capstone_amd import inferers, plan as plan_mod
from capstone_amd.models import UNet
from oracle import sliding_window as OS
from oracle.monai_unet import UNet as OracleUNet


def _pair(dims=3, chans=(4, 8, 16), strides=(2, 2), nres=2, precision="fp32", seed=3):
    torch.manual_seed(seed)
    ref = OracleUNet(dims, 1, 10, chans, strides, num_res_units=nres)
    net = UNet(dims, 1, 10, chans, strides, num_res_units=nres, precision=precision)
    net.load_state_dict(ref.state_dict())
    return ref.eval(), net


@pytest.mark.parametrize("img,roi,overlap", [((40, 36, 24), (32, 32, 16), 0.25), ((512, 512, 160), (192, 192, 64), 0.25),
                                             ((20, 64, 9), (32, 32, 16), 0.5), ((33, 17, 8), (16, 16, 8), 0.0),
                                             ((64, 64, 1), (32, 48, 1), 0.6)])
def test_window_grid_matches_monai_restatement(img, roi, overlap):
    padded = tuple(max(i, r) for i, r in zip(img, roi))
    mine = inferers._window_starts(padded, roi, inferers._scan_interval(padded, roi, overlap))
    ref = OS.dense_patch_slices(padded, roi, OS.get_scan_interval(padded, roi, overlap))
    assert mine == [tuple(s.start for s in sl) for sl in ref]
    assert all(sl[d].stop - sl[d].start == roi[d] for sl in ref for d in range(3))


@pytest.mark.parametrize("roi", [(32, 32, 16), (192, 192, 64), (16, 8, 4), (7, 9, 5)])
def test_gaussian_importance_closed_form_equals_filtered_impulse(roi):
    ref = OS.compute_importance_map(roi, "gaussian").numpy()
    mine = inferers._importance_map(roi, "gaussian", 0.125)
    np.testing.assert_allclose(mine, ref, rtol=1e-5, atol=1e-9)
    assert mine.min() > 0 and mine.max() == 1.0
    assert (inferers._importance_map(roi, "constant", 0.125) == 1).all()


@pytest.fixture()
def emu():
    from abi_emulator import Emulator, patch_native
    e = Emulator()
    undo = patch_native(nat, e)
    orig = plan_mod.Plan.run
    plan_mod.Plan.run = staticmethod(lambda prog, stream, lo=0, hi=None: e.run(prog[lo:hi]))
    yield e
    plan_mod.Plan.run = orig
    undo()


@pytest.mark.parametrize("mode,img,roi,swb", [("constant", (20, 12, 8), (8, 8, 4), 3), ("gaussian", (12, 20, 8), (8, 8, 8), 2),
                                              ("gaussian", (6, 12, 4), (8, 8, 4), 4)])
def test_inferer_host_logic_emulated(emu, mode, img, roi, swb):
    ref, net = _pair(chans=(4, 8), strides=(2,))
    x = torch.randn(1, 1, *img)
    with torch.no_grad():
        want = OS.sliding_window_inference(x, roi, swb, ref, 0.25, mode)
    got = inferers.sliding_window_inference(x, roi, swb, net, 0.25, mode)
    assert got.shape == want.shape
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-4, atol=2e-5)
    plan = net.engine().last_plan
    assert plan.inference and not plan.bwd
    with pytest.raises(RuntimeError):
        plan.backward()


def test_inferer_rejects_what_it_cannot_run(emu):
    _, net = _pair(chans=(4, 8), strides=(2,))
    with pytest.raises(NotImplementedError):
        inferers.sliding_window_inference(torch.zeros(2, 1, 8, 8, 4), (8, 8, 4), 1, net)
    with pytest.raises(TypeError):
        inferers.sliding_window_inference(torch.zeros(1, 1, 8, 8, 4), (8, 8, 4), 1, lambda t: t)
    with pytest.raises(ValueError):
        inferers.sliding_window_inference(torch.zeros(1, 1, 8, 8, 4), (7, 8, 4), 1, net)    # ROI not divisible by 2


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("mode,img,roi,swb,precision", [
    ("constant", (40, 36, 24), (32, 32, 16), 3, "fp32"),
    ("gaussian", (40, 36, 24), (32, 32, 16), 2, "fp32"),
    ("gaussian", (24, 48, 12), (32, 32, 16), 4, "fp32"),          # first/last axis smaller than the ROI: padded + cropped
    ("gaussian", (40, 36, 24), (32, 32, 16), 2, "bf16"),
    ("gaussian", (40, 36, 24), (32, 32, 16), 2, "fp16"),
    ("constant", (24, 48, 12), (32, 32, 16), 4, "fp16"),
])
def test_sliding_window_gpu_vs_oracle(mode, img, roi, swb, precision):
    ref, net = _pair(chans=(8, 16, 32), strides=(2, 2), precision=precision)
    net = net.cuda()
    x = torch.randn(1, 1, *img)
    with torch.no_grad():
        want = OS.sliding_window_inference(x, roi, swb, ref, 0.25, mode)
    got = inferers.SlidingWindowInferer(roi, swb, 0.25, mode)(x.cuda(), net).cpu()
    assert got.shape == want.shape
    if precision == "fp32":
        assert (got - want).abs().max() < 1e-3          # north_star's fp32 logit tolerance
        np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-4, atol=5e-5)
        agree = (got.argmax(1) == want.argmax(1)).float().mean().item()
        assert agree > 0.9999
    else:
        # 16-bit storage, fp32 accumulate and fp32 InstanceNorm statistics: bf16 keeps 8 significant bits, IEEE half 11
        err = (got - want).abs().max().item() / want.abs().max().item()
        assert err < (4e-2 if precision == "bf16" else 8e-3), err
        assert (got.argmax(1) == want.argmax(1)).float().mean().item() > (0.97 if precision == "bf16" else 0.995)


@pytest.mark.gpu
def test_deep_residual_unet_fp16_sliding_window_vs_oracle():
    """BASELINE.json configs[4]'s network — the 5-filter residual U-Net (32,64,128,256,512), 19.2 M parameters — in fp16 storage
    through the sliding-window inferer, against the CPU oracle on a volume the oracle finishes in seconds
    (80 x 72 x 40, ROI 64 x 64 x 32, overlap 0.25: 2 x 2 x 2 windows, Gaussian blending)."""
    ref, net = _pair(chans=(32, 64, 128, 256, 512), strides=(2, 2, 2, 2), precision="fp16", seed=5)
    net = net.cuda()
    g = torch.Generator().manual_seed(6)
    x = torch.randn(1, 1, 80, 72, 40, generator=g)
    with torch.no_grad():
        want = OS.sliding_window_inference(x, (64, 64, 32), 4, ref, 0.25, "gaussian")
    got = inferers.sliding_window_inference(x.cuda(), (64, 64, 32), 4, net, 0.25, "gaussian").cpu()
    assert net.engine().last_plan.dt == nat.F16 and net.engine().last_plan.x.t.dtype == torch.float16
    err = (got - want).abs().max().item() / want.abs().max().item()
    agree = (got.argmax(1) == want.argmax(1)).float().mean().item()
    assert err < 8e-3, err
    assert agree > 0.995, agree
    # half precision is inference-only: a training forward (grad enabled) of a precision-16 net runs in bf16 storage after ONE
    # RuntimeWarning (Lightning's --precision 16 must train: ADVICE r2) — never silently, never in fp16
    import warnings
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        net(x[:, :, :64, :64, :32].cuda())
    assert net.engine().last_plan.dt == nat.BF16 and not net.engine().last_plan.inference


@pytest.mark.gpu
def test_full_size_sliding_window_properties_fp16(monkeypatch):
    """BASELINE.json configs[4] at its stated size — 512 x 512 x 160 volume, ROI 192 x 192 x 64, overlap 0.25 (48 windows), the
    5-filter residual U-Net, fp16 — through size-independent properties:
      (1) the normalised blend weights sum to 1 at every voxel (unit predictions blend to exactly-one logits),
      (2) a rerun is bit-identical (stream-ordered accumulation, no atomics),
      (3) one output-centric blend launch per batch == one launch per window, bit for bit,
      (4) fp16 storage stays within half-precision distance of the fp32 path of the same engine on the same volume (that path
          is checked against the oracle at the sizes the oracle can run)."""
    img, roi = (512, 512, 160), (192, 192, 64)
    dev = torch.device("cuda:0")
    padded = img
    starts = inferers._window_starts(padded, roi, inferers._scan_interval(padded, roi, 0.25))
    assert len(starts) == 48
    for mode in ("constant", "gaussian"):
        imp, inv = inferers._blend_maps(dev, img, roi, (0, 0, 0), tuple(starts), mode, 0.125)
        ones = torch.ones(roi, dtype=torch.float32, device=dev)
        acc = torch.zeros(img, dtype=torch.float32, device=dev)
        for a, b, c in starts:
            nat.call("ctseg_window_blend", ones.data_ptr(), 1, 1, *roi, a, b, c, imp.data_ptr(), inv.data_ptr(), acc.data_ptr(), *img, 1)
        torch.cuda.synchronize()
        assert float((acc - 1).abs().max()) < 5e-6, mode
        del acc
    torch.manual_seed(12342)
    net = UNet(3, 1, 10, (32, 64, 128, 256, 512), (2, 2, 2, 2), num_res_units=2, precision="fp16").cuda()
    g = torch.Generator(device=dev).manual_seed(7)
    x = torch.randn(1, 1, *img, device=dev, generator=g)
    y1 = inferers.sliding_window_inference(x, roi, 4, net, 0.25, "gaussian").clone()
    y2 = inferers.sliding_window_inference(x, roi, 4, net, 0.25, "gaussian").clone()
    assert torch.isfinite(y1).all()
    assert torch.equal(y1, y2)                                                          # (2)
    monkeypatch.setenv("CTSEG_SW_BLEND", "per_window")
    y3 = inferers.sliding_window_inference(x, roi, 4, net, 0.25, "gaussian").clone()
    monkeypatch.delenv("CTSEG_SW_BLEND")
    assert torch.equal(y1, y3)                                                          # (3)
    del y2, y3
    net32 = UNet(3, 1, 10, (32, 64, 128, 256, 512), (2, 2, 2, 2), num_res_units=2, precision="fp32")
    net32.load_state_dict(net.state_dict())
    net32 = net32.cuda()
    monkeypatch.setenv("CTSEG_SW_DEVICE_BATCH", "8")
    y32 = inferers.sliding_window_inference(x, roi, 4, net32, 0.25, "gaussian")
    err = float((y1 - y32).abs().max() / y32.abs().max())
    agree = float((y1.argmax(1) == y32.argmax(1)).float().mean())
    assert err < 8e-3, err                                                              # (4)
    assert agree > 0.995, agree


@pytest.mark.gpu
def test_sliding_window_2d_and_plan_reuse():
    ref, net = _pair(dims=2, chans=(8, 16), strides=(2,))
    net = net.cuda()
    x = torch.randn(1, 1, 40, 28)
    with torch.no_grad():
        want = OS.sliding_window_inference(x, (16, 16), 4, ref, 0.5, "gaussian")
    got = inferers.sliding_window_inference(x.cuda(), (16, 16), 4, net, 0.5, "gaussian")
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=1e-4, atol=5e-5)
    n_plans = len(net.engine().plans)
    inferers.sliding_window_inference(x.cuda(), (16, 16), 4, net, 0.5, "constant")
    assert len(net.engine().plans) == n_plans           # the inference plan is cached per (batch, ROI)


@pytest.mark.gpu
def test_device_window_batch_changes_nothing_beyond_partial_sum_order(monkeypatch):
    """the forward batch on the device may be larger than the caller's sw_batch_size (windows are independent samples, blended
    in scan order either way): same logits up to the order of the fp32 InstanceNorm partial sums, same argmax"""
    ref, net = _pair(chans=(8, 16, 32), strides=(2, 2), precision="fp32")
    net = net.cuda()
    x = torch.randn(1, 1, 40, 36, 24).cuda()
    monkeypatch.setenv("CTSEG_SW_DEVICE_BATCH", "0")
    exact = inferers.sliding_window_inference(x, (32, 32, 16), 2, net, 0.25, "gaussian").clone()
    assert inferers.LAST_DEVICE_BATCH == 2
    monkeypatch.setenv("CTSEG_SW_DEVICE_BATCH", "auto")
    auto = inferers.sliding_window_inference(x, (32, 32, 16), 2, net, 0.25, "gaussian")
    assert inferers.LAST_DEVICE_BATCH == 8          # all 8 windows of this volume in one forward
    assert (auto - exact).abs().max().item() < 2e-5 * exact.abs().max().item()
    assert torch.equal(auto.argmax(1), exact.argmax(1))


@pytest.mark.gpu
def test_batched_blend_equals_per_window_blend_bit_for_bit():
    """ctseg_window_blend_batch (one output-centric launch) == the per-window ctseg_window_blend calls in the same order"""
    from capstone_amd import _native as nat
    g = torch.Generator().manual_seed(41)
    X, Y, Z, C, ld = 20, 18, 14, 10, 12
    roi = (12, 10, 8)
    starts = [(0, 0, 0), (8, 0, 0), (0, 8, 6), (8, 8, 6), (4, 4, 3), (-2, 3, 0), (10, 9, 7)]     # overlapping, one hanging off the volume
    nw = len(starts)
    logits = torch.randn(nw, roi[0] * roi[1] * roi[2], ld, generator=g).cuda()
    imp = (torch.rand(roi, generator=g) + 0.1).cuda()
    inv = (torch.rand(X, Y, Z, generator=g) + 0.5).cuda()
    a = torch.zeros(X, Y, Z, ld, device="cuda")
    b = torch.zeros_like(a)
    for w, (x0, y0, z0) in enumerate(starts):
        nat.call("ctseg_window_blend", logits[w].data_ptr(), ld, C, *roi, x0, y0, z0, imp.data_ptr(), inv.data_ptr(), a.data_ptr(),
                 X, Y, Z, ld)
    st = torch.tensor(starts, dtype=torch.int32).cuda()
    nat.call("ctseg_window_blend_batch", logits.data_ptr(), ld, C, *roi, st.data_ptr(), nw, imp.data_ptr(), inv.data_ptr(),
             b.data_ptr(), X, Y, Z, ld, None)
    torch.cuda.synchronize()
    assert torch.equal(a[..., :C], b[..., :C])
    # a second batch accumulates on top of the first, as successive forward batches of one volume do
    for w, (x0, y0, z0) in enumerate(starts[:3]):
        nat.call("ctseg_window_blend", logits[w].data_ptr(), ld, C, *roi, x0, y0, z0, imp.data_ptr(), inv.data_ptr(), a.data_ptr(),
                 X, Y, Z, ld)
    import ctypes
    bbox = (ctypes.c_int32 * 6)(0, 0, 0, 20, 18, 14)      # windows 0..2 span x 0..19, y 0..17, z 0..13
    nat.call("ctseg_window_blend_batch", logits.data_ptr(), ld, C, *roi, st.data_ptr(), 3, imp.data_ptr(), inv.data_ptr(),
             b.data_ptr(), X, Y, Z, ld, bbox)
    torch.cuda.synchronize()
    assert torch.equal(a[..., :C], b[..., :C])
