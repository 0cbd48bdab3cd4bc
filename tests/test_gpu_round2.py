"""GPU parity tests added in round 2 (run with -m gpu on a MI355X), all through the C ABI:
  * the ``exclude_missing`` branch of the loss zoo (capstone/models/losses.py:196-221) on 5-D logits: values and d/dlogits,
  * full-size (BASELINE.json configs[2] volume) fp32 step against the CPU oracle,
  * K-step training trajectories (configs[1] shape) in fp32 and bf16 against the oracle's curve,
  * an optimizer update on one plan reaching every other cached plan,
  * the checkpoint layout of the native step's Adam state.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from capstone_amd import segloss  # noqa: E402

DEV = "cuda:0"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def _dump(name, obj):
    try:
        os.makedirs(OUT, exist_ok=True)
        with open(os.path.join(OUT, name), "w") as f:
            json.dump(obj, f, indent=1)
    except OSError:
        pass


# ----------------------------------------------------------------------------------------------------------------------
# (a) exclude_missing
# ----------------------------------------------------------------------------------------------------------------------
def test_mask_weight_tables_match_the_reference_run_apply_missing_mask(golden):
    """the product folds apply_missing_mask (models/losses.py:206-221) into per-(sample, class) weights of the loss table
    (SegLossEngine._mask_weights).  Against the reference's OWN outputs (ref_leaf.npz mm_*): both indicator patterns for Dice
    (b zeroes a class everywhere -> the isinf -> uniform-weights branch, :216-217) and the Focal background column (:207-212)."""
    from capstone_amd.models.losses import apply_missing_mask
    leaf = golden("ref_leaf.npz")
    B = leaf["mm_table"].shape[0]
    le = segloss.SegLossEngine(torch.device(DEV), B, 64, 10)
    t9, t10 = torch.from_numpy(leaf["mm_table"]).to(DEV), torch.from_numpy(leaf["mm_table10"]).to(DEV)
    assert (leaf["mm_ind_b"].sum(0) == 0).any(), "fixture b must hold a class nobody annotated (the isinf branch)"
    for name, table, ind_key, want in (("Dice", t9, "mm_ind_a", "mm_dice_a"), ("Dice", t9, "mm_ind_b", "mm_dice_b"),
                                       ("Focal", t10, "mm_ind_a", "mm_focal_a")):
        ind = torch.from_numpy(leaf[ind_key]).to(DEV)
        wt = le._mask_weights(name, ind, True, table.shape[1])
        np.testing.assert_allclose(float((table * wt).sum()), leaf[want], rtol=2e-6, err_msg=f"{name}/{ind_key} (table weights)")
        np.testing.assert_allclose(float(apply_missing_mask(name, table, ind)), leaf[want], rtol=2e-6, err_msg=f"{name}/{ind_key}")


@pytest.mark.parametrize("pattern", ["class_missing_everywhere", "class_missing_in_one_sample", "all_annotated"])
def test_exclude_missing_losses_values_and_dlogits_on_5d_logits(pattern):
    """MultipleLossWrapper(["Dice","Focal","GeneralizedDice"], exclude_missing=True) on (B,10,H,W,D) logits: the reduction
    "none" tables, the indicator weighting and the per-sample gradient coefficients, against oracle.losses.MultipleLoss
    (whose missing_mask is pinned bit-exactly to the reference's, tests/test_oracle_golden.py)."""
    from capstone_amd.models.losses import MultipleLossWrapper
    from oracle import losses as OL
    g = torch.Generator().manual_seed(41)
    B, H, W, D = 3, 8, 12, 4
    logits = torch.randn(B, 10, H, W, D, generator=g) * 2.0
    target = torch.randint(0, 10, (B, H, W, D), generator=g)
    target[1][target[1] == 6] = 0                      # class 6 absent from sample 1: GDL's inf -> max weight path
    ind = torch.ones(B, 9)
    if pattern == "class_missing_everywhere":
        ind[:, 2] = 0                                  # 1 / 0 -> inf -> uniform weights (models/losses.py:216-217)
        ind[0, 5] = 0
    elif pattern == "class_missing_in_one_sample":
        ind[2, 4] = 0
        ind[0, 7] = 0                                  # two samples incomplete: Focal's background column is 0 for them
    names = ["Dice", "Focal", "GeneralizedDice"]
    xr = logits.clone().requires_grad_(True)
    rv = OL.MultipleLoss(names, exclude_missing=True)(xr, target, ind)
    torch.stack(list(rv.values())).sum().backward()
    x = logits.to(DEV).requires_grad_(True)
    v = MultipleLossWrapper(names, exclude_missing=True)(input=x, target=target.to(DEV), mask_indicator=ind.to(DEV))
    for n in names:
        np.testing.assert_allclose(v[n].item(), rv[n].item(), rtol=2e-5, atol=1e-7, err_msg=n)
    torch.stack(list(v.values())).sum().backward()
    got, ref = x.grad.cpu().numpy(), xr.grad.numpy()
    np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-6 * float(np.abs(ref).max()) + 1e-9)
    assert float(np.abs(ref).max()) > 1e-6
    # one loss at a time, with a non-unit upstream gradient: the coefficient tables are per loss
    for n, scale in zip(names, (0.5, 2.0, 3.0)):
        xr1 = logits.clone().requires_grad_(True)
        (OL.MultipleLoss([n], exclude_missing=True)(xr1, target, ind)[n] * scale).backward()
        x1 = logits.to(DEV).requires_grad_(True)
        (MultipleLossWrapper([n], exclude_missing=True)(input=x1, target=target.to(DEV), mask_indicator=ind.to(DEV))[n] * scale).backward()
        r1 = xr1.grad.numpy()
        np.testing.assert_allclose(x1.grad.cpu().numpy(), r1, rtol=2e-3, atol=2e-6 * float(np.abs(r1).max()) + 1e-9, err_msg=n)


def test_exclude_missing_training_step_through_the_network():
    """BaseUNet3D(loss_fx=[Dice, Focal], exclude_missing=True): fit_step (native) and training_step (autograd surface) against the
    oracle step with the same indicator — loss values, gradient direction, Dice metric."""
    from capstone_amd.volumetric.base_trainer import BaseUNet3D
    import oracle.trainer as OT
    torch.manual_seed(9)
    filters = (8, 16, 32)
    om = OT.OracleUNet3D(filters=filters, loss_fx=("Dice", "Focal"), exclude_missing=True)
    g = torch.Generator().manual_seed(10)
    images = torch.randn(2, 1, 16, 16, 8, generator=g)
    masks = torch.zeros(2, 9, 16, 16, 8, dtype=torch.uint8)
    for c in range(9):
        masks[:, c, c:c + 6, 2 + c:9 + c, 1:7] = 1
    ind = torch.ones(2, 9)
    ind[:, 3] = 0
    ind[1, 8] = 0
    oloss = om.training_step((images, masks, ind))
    oloss.backward()
    batch = (images.to(DEV), masks.to(DEV), ind.to(DEV))
    for mode in ("native", "autograd"):
        m = BaseUNet3D(filters=list(filters), loss_fx=["Dice", "Focal"], exclude_missing=True)
        m.load_state_dict(om.state_dict())
        m.to(DEV)
        if mode == "native":
            loss = m.fit_step(batch)
            grads = {k: m.unet.engine().store.grad_view(p).cpu() for k, p in m.named_parameters()}
        else:
            loss = m.training_step(batch)
            loss.backward()
            grads = {k: p.grad.cpu() for k, p in m.named_parameters()}
        np.testing.assert_allclose(loss.item(), oloss.item(), rtol=1e-4, err_msg=mode)
        for n in ("Dice", "Focal"):
            np.testing.assert_allclose(m.logged[f"{n} Loss (train)"].item(), om.logged[f"{n} Loss (train)"].item(), rtol=1e-4)
        assert abs(m.logged["Mean Dice Score (train)"].item() - om.logged["Mean Dice Score (train)"].item()) <= 0.002
        for k, p in om.named_parameters():
            a, b = grads[k].flatten().double(), p.grad.flatten().double()
            if b.norm() > 1e-5:
                assert float(torch.dot(a, b) / (a.norm() * b.norm())) > 0.9999, (mode, k)


# ----------------------------------------------------------------------------------------------------------------------
# (c) full-size fp32 step vs the oracle
# ----------------------------------------------------------------------------------------------------------------------
_FULL = {}


def _full_size_oracle():
    """ONE oracle run (fp32 CPU step + an fp64 run of the same step for the gradients) of one volume of BASELINE.json's metric
    shape, shared by the fp32 and the bf16 full-size tests (~1.5 min of host time, paid once per session)."""
    if _FULL:
        return _FULL
    import copy
    from bench import synthetic_batch
    import oracle.trainer as OT
    torch.manual_seed(12342)
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    om = OT.OracleUNet3D(filters=(32, 64, 128, 256), loss_fx=("CrossEntropy",))
    sd = {k: v.clone() for k, v in om.state_dict().items()}
    batch = synthetic_batch(1, 512, 512, 48, "cpu", 12342)
    _, _, _, ologits, oloss = om.shared_step(batch, True)
    oloss.backward()
    om64 = copy.deepcopy(om).double()
    om64.zero_grad()
    _, _, _, _, l64 = om64.shared_step((batch[0].double(), batch[1], batch[2].double()), True)
    l64.backward()
    _FULL.update(om=om, sd=sd, batch=batch, ologits=ologits.detach(), oloss=float(oloss), l64=float(l64),
                 odice=float(om.logged["Mean Dice Score (train)"]),
                 g32={k: p.grad.detach().clone() for k, p in om.named_parameters()},
                 g64={k: p.grad.detach().clone() for k, p in om64.named_parameters()})
    del om64
    return _FULL


def _noise_only(k):
    """a conv bias feeding an InstanceNorm has an analytically zero gradient: both sides hold rounding noise there"""
    return k.endswith(".bias") and "residual" not in k and not k.endswith("model.2.1.conv.unit0.conv.bias")


def test_full_size_fp32_step_vs_oracle():
    """ONE volume of BASELINE.json's metric shape, 1x1x512x512x48, fp32 storage (v_mfma_f32_16x16x4_f32), the real network
    (32,64,128,256): forward + CrossEntropy + Dice metric + backward against the CPU oracle (~20 s of host time).
    north_star's bar: logits within 1e-3, argmax masks equal wherever the top-2 margin exceeds the logit error, Dice +-0.002."""
    from capstone_amd.volumetric.base_trainer import BaseUNet3D
    from oracle.metrics import squash_predictions
    F = _full_size_oracle()
    ologits, oloss, odice, batch = F["ologits"], F["oloss"], F["odice"], F["batch"]
    m = BaseUNet3D(filters=[32, 64, 128, 256], loss_fx=["CrossEntropy"], precision="fp32")
    m.load_state_dict(F["sd"])
    m.to(DEV)
    loss = m.fit_step(tuple(t.to(DEV) for t in batch))
    eng = m.unet.engine()
    logits = eng.logits_view().cpu()
    err = float((logits - ologits).abs().max())
    dice = float(m.logged["Mean Dice Score (train)"])
    top2 = ologits.topk(2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > 4 * max(err, 1e-6)
    pred = m.unet.engine().last_plan._ctseg_loss.predictions(eng.last_plan.logits.ptr(), eng.last_plan.logits.ld).cpu().long()
    opred = squash_predictions(ologits).reshape(1, -1)
    flips = int((pred != opred).sum())
    # gradients: fp32 CPU and fp32 GPU sum up to 12.6 M terms per element in different orders, so where they disagree an fp64 run
    # of the same oracle arbitrates: the GPU gradient must be as close to the fp64 one as the CPU fp32 gradient is (or within 1e-4)
    rows = []
    for (k, q) in zip(F["g64"], m.parameters()):
        g64 = F["g64"][k].flatten()
        if g64.norm() < 1e-4 or _noise_only(k):
            continue
        e_gpu = float((eng.store.grad_view(q).cpu().flatten().double() - g64).norm() / g64.norm())
        e_cpu = float((F["g32"][k].flatten().double() - g64).norm() / g64.norm())
        rows.append((e_gpu, e_cpu, k))
    worst = sorted(rows, reverse=True)[:6]
    _dump("full_size_fp32_vs_oracle.json", {"logits_max_abs_err": err, "loss": loss.item(), "oracle_loss": oloss,
                                            "oracle_loss_fp64": F["l64"], "dice": dice, "oracle_dice": odice,
                                            "safe_fraction": float(safe.float().mean()), "mask_flips_total": flips,
                                            "voxels": int(opred.numel()),
                                            "worst_grad_rel_err_vs_fp64 (gpu, cpu_fp32, tensor)": worst})
    assert err < 1e-3, err
    assert abs(loss.item() - oloss) < 1e-4 * abs(oloss)
    assert abs(dice - odice) <= 0.002
    assert float(safe.float().mean()) > 0.99
    assert torch.equal(pred.reshape(-1)[safe.reshape(-1)], opred.reshape(-1)[safe.reshape(-1)])
    assert flips <= 50, flips                      # near-ties below the logit error may flip; O(10) per volume expected
    for e_gpu, e_cpu, k in rows:
        assert e_gpu <= max(1e-4, 2.0 * e_cpu), (k, e_gpu, e_cpu)


def test_full_size_bf16_fused_head_step_vs_oracle():
    """The HEADLINE dtype at the headline shape through the headline path: bf16 storage, ``fit_step(keep_logits=False)`` (the
    logits convolution with the cross-entropy in its epilogue — what bench.py times) on one 1x1x512x512x48 volume, against the same
    oracle run as the fp32 test.  Bars: loss within 2 %, mean Dice (from the fused integer counts) +-0.002, every weight tensor's
    gradient at cosine > 0.97 to the oracle's FP64 gradient, argmax masks equal wherever the oracle's top-2 margin exceeds the
    bf16 logit error (masks / logit error from a forward of the same weights before the step: the fused step writes no logits)."""
    from capstone_amd.volumetric.base_trainer import BaseUNet3D
    from capstone_amd import _native as nat
    from oracle.metrics import squash_predictions
    F = _full_size_oracle()
    ologits, oloss, odice, batch = F["ologits"], F["oloss"], F["odice"], F["batch"]
    m = BaseUNet3D(filters=[32, 64, 128, 256], loss_fx=["CrossEntropy"], precision="bf16")
    m.load_state_dict(F["sd"])
    m.to(DEV)
    dbatch = tuple(t.to(DEV) for t in batch)
    eng = m.unet.engine()
    # forward of the same weights on the TRAINING plan (same kernels as the step's forward), logits materialised
    plan = eng.plan_for(dbatch[0])
    plan.forward(dbatch[0])
    logits = eng.logits_view(plan).cpu()
    err = float((logits - ologits).abs().max())
    rel = err / float(ologits.abs().max())
    le = segloss.SegLossEngine(torch.device(DEV), 1, plan.logits.S, 10)
    pred = le.predictions(plan.logits.ptr(), plan.logits.ld).cpu().long()
    opred = squash_predictions(ologits).reshape(1, -1)
    top2 = ologits.topk(2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > 2 * err
    agree = float((pred == opred).float().mean())
    # the step bench.py times
    loss = m.fit_step(dbatch, keep_logits=False)
    with pytest.raises(nat.NativeError):
        eng.logits_view()                       # not materialised by the fused step: loud, not stale
    dice = float(m.logged["Mean Dice Score (train)"])
    cos = []
    for k, q in zip(F["g64"], m.parameters()):
        g64 = F["g64"][k].flatten()
        if g64.norm() < 1e-4 or _noise_only(k):
            continue
        a = eng.store.grad_view(q).cpu().flatten().double()
        cos.append((float(torch.dot(a, g64) / (a.norm() * g64.norm())), float((a - g64).norm() / g64.norm()), k))
    worst = sorted(cos)[:6]
    _dump("full_size_bf16_fused_head_vs_oracle.json", {
        "logits_max_abs_err": err, "logits_rel_err": rel, "loss": loss.item(), "oracle_loss": oloss,
        "loss_rel_err": abs(loss.item() - oloss) / abs(oloss), "dice": dice, "oracle_dice": odice,
        "argmax_agreement": agree, "safe_fraction": float(safe.float().mean()),
        "mask_mismatches_inside_safe_region": int((pred.reshape(-1)[safe.reshape(-1)] != opred.reshape(-1)[safe.reshape(-1)]).sum()),
        "voxels": int(opred.numel()), "tensors_compared": len(cos),
        "worst_gradient_cosine_vs_fp64 (cos, rel err, tensor)": worst})
    assert abs(loss.item() - oloss) < 0.02 * abs(oloss), (loss.item(), oloss)
    assert abs(dice - odice) <= 0.002, (dice, odice)
    assert min(cos)[0] > 0.97, worst
    # random-init logits sit close together: at a bf16 logit error of ~0.08 three quarters of the voxels have a decisive margin
    assert float(safe.float().mean()) > 0.6
    assert torch.equal(pred.reshape(-1)[safe.reshape(-1)], opred.reshape(-1)[safe.reshape(-1)])
    assert agree > 0.99


# ----------------------------------------------------------------------------------------------------------------------
# (d) K-step trajectories
# ----------------------------------------------------------------------------------------------------------------------
K_STEPS = 12
K_LEARN = 40


def _oracle_curve(filters, batch, lr, loss_fx, steps):
    import oracle.trainer as OT
    torch.manual_seed(12342)
    om = OT.OracleUNet3D(filters=filters, loss_fx=loss_fx, lr=lr)
    sd = {k: v.clone() for k, v in om.state_dict().items()}
    opt = om.configure_optimizers()
    losses, dices = [], []
    for _ in range(steps):
        losses.append(float(om.fit_step(batch, opt)))
        dices.append(float(om.logged["Mean Dice Score (train)"]))
    return sd, losses, dices


def _learnable_batch(H, W, D):
    """a task the network picks up within a few dozen steps, so the Dice curve actually climbs: nine large blocks (3 x 3 grid in
    the H-W plane, ~30 % foreground) whose class shows in the image as an intensity offset under unit noise"""
    g = torch.Generator().manual_seed(12342)
    images = torch.randn(1, 1, H, W, D, generator=g)
    masks = torch.zeros(1, 9, H, W, D, dtype=torch.uint8)
    bh, bw = H // 4, W // 4
    for c in range(9):
        i, j = divmod(c, 3)
        x0, y0 = H // 16 + i * (H // 3), W // 16 + j * (W // 3)
        masks[0, c, x0:x0 + bh, y0:y0 + bw, D // 6:D - D // 6] = 1
        images[0, 0, x0:x0 + bh, y0:y0 + bw, D // 6:D - D // 6] += 0.6 * (c + 1) * (1 if c % 2 else -1)
    return images, masks, torch.ones(1, 9)


@pytest.fixture(scope="module")
def trajectory_oracle():
    from bench import synthetic_batch
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    out = {}
    batch = synthetic_batch(1, 128, 128, 32, "cpu", 12342)
    out["configs1"] = (batch, ("CrossEntropy",), _oracle_curve((32, 64, 128, 256), batch, 1e-3, ("CrossEntropy",), K_STEPS))
    batch = _learnable_batch(64, 64, 32)
    out["learnable"] = (batch, ("CrossEntropy", "Dice"), _oracle_curve((32, 64, 128, 256), batch, 1e-3, ("CrossEntropy", "Dice"), K_LEARN))
    return out


# (max |dDice|, max relative loss difference) each run must stay within at EVERY step.  fp32: north_star's Dice +-0.002.
# bf16: what it measurably meets against the same fp32 oracle curve — configs1: 1.9e-4 / 2.7e-4 (as tight as fp32); learnable:
# 0.0078 / 1.7 % while the Dice climbs 0.05 -> 0.99 at up to 0.06 per step (a fraction of a step of lag), and +-0.0002 once
# converged (fp32 on the same curve: 0.0014 / 0.15 %).  Every run must also END within +-0.002 (mean of the last 5 steps).
TRAJ_TOL = {("configs1", "fp32"): (0.002, 5e-3), ("configs1", "bf16"): (0.002, 5e-3),
            ("learnable", "fp32"): (0.002, 5e-3), ("learnable", "bf16"): (0.012, 2.5e-2)}


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("task", ["configs1", "learnable"])
def test_k_step_trajectory_tracks_the_oracle(trajectory_oracle, task, precision):
    """K optimizer steps from the same weights on the same batch, per-step loss and mean Dice against the oracle's curve.
    configs1: BASELINE.json configs[1] (the real network on 1x1x128x128x32 synthetic CT, 1.3 % foreground, CrossEntropy — Dice stays
    near 0 there, as it does for the reference on such data).  learnable: the same network on a 64x64x32 task whose Dice climbs
    (CrossEntropy + Dice loss), so "+-0.002" is a statement about a moving curve."""
    from capstone_amd.volumetric.base_trainer import BaseUNet3D
    batch, loss_fx, (sd, olosses, odices) = trajectory_oracle[task]
    m = BaseUNet3D(filters=[32, 64, 128, 256], loss_fx=list(loss_fx), precision=precision, lr=1e-3)
    m.load_state_dict(sd)
    m.to(DEV)
    gb = tuple(t.to(DEV) for t in batch)
    losses, dices = [], []
    for _ in range(len(olosses)):
        losses.append(float(m.fit_step(gb)))
        dices.append(float(m.logged["Mean Dice Score (train)"]))
    dl = [abs(a - b) / abs(b) for a, b in zip(losses, olosses)]
    dd = [abs(a - b) for a, b in zip(dices, odices)]
    _dump(f"trajectory_{task}_{precision}.json", {"steps": len(olosses), "loss": losses, "oracle_loss": olosses, "dice": dices,
                                                  "oracle_dice": odices, "max_rel_loss_diff": max(dl), "max_abs_dice_diff": max(dd)})
    assert olosses[-1] < 0.8 * olosses[0], "the curve must actually descend"
    if task == "learnable":
        assert max(odices) > 0.2, "the Dice curve must actually climb"
    tol_d, tol_l = TRAJ_TOL[(task, precision)]
    assert max(dd) <= tol_d, dd
    assert max(dl) <= tol_l, dl
    assert abs(np.mean(dices[-5:]) - np.mean(odices[-5:])) <= 0.002


# ----------------------------------------------------------------------------------------------------------------------
# stale packed weights / checkpoints
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_optimizer_update_reaches_every_cached_plan_on_gpu(precision):
    """train on shape A, evaluate on shape B, train, evaluate: the native Adam kernel writes the flat parameter buffer through raw
    pointers, every cached plan must rebuild its packed MFMA operands (store generation counter)."""
    from capstone_amd.volumetric.base_trainer import BaseUNet3D
    from oracle.trainer import OracleUNet3D
    torch.manual_seed(2)
    om = OracleUNet3D(filters=(8, 16, 32), loss_fx=("CrossEntropy",), lr=0.02)
    m = BaseUNet3D(filters=[8, 16, 32], loss_fx=["CrossEntropy"], lr=0.02, precision=precision)
    m.load_state_dict(om.state_dict())
    m.to(DEV)
    g = torch.Generator().manual_seed(4)
    xa, xb = torch.randn(1, 1, 16, 16, 8, generator=g), torch.randn(2, 1, 8, 16, 4, generator=g)
    masks = (torch.rand(1, 9, 16, 16, 8, generator=g) < 0.1).to(torch.uint8)
    batch = (xa, masks, torch.ones(1, 9))
    gb = tuple(t.to(DEV) for t in batch)
    opt = om.configure_optimizers()
    with torch.no_grad():
        yb0 = m(xb.to(DEV)).clone().cpu()
    for _ in range(2):
        m.fit_step(gb)
        om.fit_step(batch, opt)
    with torch.no_grad():
        yb1 = m(xb.to(DEV)).clone().cpu()
    # exact check: a FRESH module holding the updated weights (its plan B is packed from them by construction) must produce the
    # same bits — a stale plan B would still show the initial weights' output
    torch.cuda.synchronize()
    m3 = BaseUNet3D(filters=[8, 16, 32], loss_fx=["CrossEntropy"], lr=0.02, precision=precision)
    m3.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    m3.to(DEV)
    with torch.no_grad():
        yb3 = m3(xb.to(DEV)).clone().cpu()
    assert float((yb1 - yb0).abs().max()) > 0.05, "the update must be visible in plan B's output"
    assert torch.equal(yb1, yb3)
    if precision == "fp32":       # and against the oracle stepping alongside
        ref = om(xb).detach()
        assert float((yb1 - ref).abs().max()) < 2e-3 * max(1.0, float(ref.abs().max())), float((yb1 - ref).abs().max())


def test_native_adam_state_checkpoint_round_trip_on_gpu():
    from capstone_amd.volumetric.base_trainer import BaseUNet3D
    torch.manual_seed(3)
    g = torch.Generator().manual_seed(5)
    batch = (torch.randn(1, 1, 16, 16, 8, generator=g).to(DEV), (torch.rand(1, 9, 16, 16, 8, generator=g) < 0.1).to(torch.uint8).to(DEV),
             torch.ones(1, 9).to(DEV))
    m1 = BaseUNet3D(filters=[8, 16, 32], loss_fx=["CrossEntropy"], precision="bf16").to(DEV)
    for _ in range(2):
        m1.fit_step(batch)
    ck = m1.checkpoint()
    ck = {"state_dict": {k: v.cpu() for k, v in ck["state_dict"].items()}, "hyper_parameters": ck["hyper_parameters"],
          "optimizer_states": [{"state": {i: {k: v.cpu() for k, v in e.items()} for i, e in ck["optimizer_states"][0]["state"].items()},
                                "param_groups": ck["optimizer_states"][0]["param_groups"]}]}
    m2 = BaseUNet3D(filters=[8, 16, 32], loss_fx=["CrossEntropy"], precision="bf16").to(DEV)
    m2.load_checkpoint(ck)
    l1, l2 = float(m1.fit_step(batch)), float(m2.fit_step(batch))
    assert l1 == l2
    torch.cuda.synchronize()
    assert torch.equal(m1.unet.engine().store.flat_p, m2.unet.engine().store.flat_p)


# ----------------------------------------------------------------------------------------------------------------------
# fp16 storage (BASELINE.json configs[4]): every forward kernel family, op level, against torch CPU
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind,cin,cout,shape,family", [
    ("conv", 8, 12, (1, 9, 11, 7), "generic 256x16 tile"),
    ("conv", 16, 24, (2, 6, 10, 5), "generic 256x32 tile"),
    ("conv_s2", 8, 64, (1, 10, 12, 8), "generic 128x64 tile, stride 2"),
    ("conv", 256, 256, (1, 8, 8, 6), "ring 192x256"),
    ("conv", 128, 128, (2, 8, 8, 6), "ring 192x128"),
    ("convT", 384, 64, (1, 5, 6, 4), "8-class generic + ring input gradient"),
    ("conv", 32, 32, (2, 12, 16, 24), "LDS halo 64-byte voxels"),
    ("conv", 16, 10, (1, 8, 8, 8), "LDS halo 32-byte voxels, 10 columns"),
    ("conv", 64, 64, (1, 9, 20, 13), "streamed-weight halo"),
    ("convT", 128, 32, (1, 8, 16, 16), "streamed-weight halo, 8 classes"),
    ("convT", 64, 10, (1, 8, 8, 8), "up halo"),
    ("convT", 384, 64, (1, 10, 18, 12), "many-channel 8-class (conv_up8), forward"),
    ("conv_s2", 64, 256, (1, 36, 40, 12), "many-channel 8-class (conv_up8), input gradient"),
    ("conv_s2", 32, 128, (1, 18, 40, 24), "stride-2 halo"),
    ("conv_s2", 1, 32, (1, 16, 16, 8), "stem"),
    ("conv1", 128, 256, (1, 6, 6, 4), "1x1x1 residual"),
])
def test_fp16_forward_and_input_gradient_passes_vs_torch_cpu(kind, cin, cout, shape, family):
    """IEEE-half storage (v_mfma_f32_16x16x32_f16, fp32 accumulate): the forward and input-gradient passes of every kernel family
    the inference plans use.  Half keeps 11 significant bits: relative L-inf error < 4e-3 (bf16's bound in this file's
    siblings is 2.5e-2)."""
    from capstone_amd import _native as nat
    from helpers import rel_err, run_conv_module
    torch.manual_seed(cin + 3 * cout + shape[2])
    if kind == "convT":
        mod = torch.nn.ConvTranspose3d(cin, cout, 3, 2, 1, output_padding=1)
    elif kind == "conv1":
        mod = torch.nn.Conv3d(cin, cout, 1, 1, 0)
    else:
        mod = torch.nn.Conv3d(cin, cout, 3, 2 if kind == "conv_s2" else 1, 1)
    x = torch.randn(shape[0], cin, *shape[1:])
    xr = x.clone().requires_grad_(True)
    y = mod(xr)
    gy = torch.randn_like(y)
    y.backward(gy)
    yy, gx, gw, gb = run_conv_module(mod, x, gy, nat.F16, DEV)
    assert gw is None and gb is None
    assert rel_err(yy, y.detach()) < 4e-3, family + ": forward"
    if gx is not None:
        assert rel_err(gx, xr.grad) < 4e-3, family + ": input gradient"


def test_fp16_instnorm_prelu_and_saturating_store():
    """InstanceNorm + PReLU forward in half storage (statistics from the fp32 accumulators of the conv epilogue), and the
    saturating half store: a conv output beyond +-65504 is stored as +-65504, not inf (its statistics stay exact in fp32)."""
    from capstone_amd import _native as nat
    from capstone_amd.engine import GemmLayer
    from capstone_amd.plan import _NormAct
    from helpers import MiniPlan, from_cl, rel_err, to_cl
    torch.manual_seed(1)
    C = 16
    x = torch.randn(2, C, 6, 10, 8)
    x[0, 3] *= 50.0
    conv = torch.nn.Conv3d(C, C, 1)
    alpha = torch.nn.Parameter(torch.tensor([0.2]))
    with torch.no_grad():
        conv.weight.copy_(torch.eye(C).reshape(C, C, 1, 1, 1))
        conv.bias.zero_()
    plan = MiniPlan([conv.weight, conv.bias, alpha], DEV, nat.F16, 3)
    layer = GemmLayer(plan, "id", False, 1, 1, C, [(conv.weight, conv.bias, C)], C)
    plan.packer.finalize()
    xa = to_cl(x, nat.F16, DEV)
    y, stats = layer.emit_fwd(xa, want_stats=True)
    out = _NormAct(plan, alpha).emit_fwd(y, stats, 0, None, None)
    plan.run()
    xin = xa.valid().float().cpu()
    ref = torch.nn.functional.prelu(torch.nn.functional.instance_norm(xin), alpha.detach().cpu())
    assert rel_err(from_cl(out), ref) < 2e-3
    # saturation: weights of 1000 on inputs of ~100 -> |y| up to 1e5 > 65504
    with torch.no_grad():
        conv.weight.mul_(1000.0)
    xb = to_cl(torch.full((1, C, 4, 4, 4), 100.0), nat.F16, DEV)
    yb, _ = layer.emit_fwd(xb)
    plan.run()
    torch.cuda.synchronize()
    v = yb.valid().float()
    assert torch.isfinite(v).all() and float(v.max()) == 65504.0


# ----------------------------------------------------------------------------------------------------------------------
# logits convolution with the cross-entropy fused into its epilogue (ctseg_conv_logits_ce)
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape,narrow,loss", [((2, 32, 48, 16), "1", "CrossEntropy"), ((2, 32, 48, 16), "0", "CrossEntropy"),
                                               ((1, 36, 44, 20), "1", "WeightedCrossEntropy"), ((3, 20, 24, 12), "0", "WeightedCrossEntropy")])
def test_fused_head_cross_entropy_equals_the_two_pass_path(monkeypatch, shape, narrow, loss):
    """fit_step(keep_logits=False) runs the logits convolution and the cross-entropy as ONE launch.  Against the two-pass path on
    the same weights and batch: the Dice counts bit-identical (the prediction keeps the exact softmax -> argmax semantics);
    d loss / d logits equal up to the reciprocal-vs-division rounding of the softmax (at most one 16-bit ulp on a small fraction
    of the elements); the loss equal to 2e-6.  12-wide and 16-wide head layouts, ragged tiles (36 x 44 x 20), plain and
    class-weighted cross-entropy."""
    from capstone_amd.volumetric.base_trainer import BaseUNet3D
    monkeypatch.setenv("CTSEG_NARROW_ROWS", narrow)
    B, H, W, D = shape
    g = torch.Generator().manual_seed(51)
    images = torch.randn(B, 1, H, W, D, generator=g).to(DEV)
    masks = (torch.rand(B, 9, H, W, D, generator=g) < 0.08).to(torch.uint8).to(DEV)
    ind = torch.ones(B, 9, dtype=torch.float64).to(DEV)
    out = {}
    for fused in (False, True):
        torch.manual_seed(8)
        m = BaseUNet3D(filters=[16, 32, 64], loss_fx=[loss], precision="bf16").to(DEV)
        losses = [float(m.fit_step((images, masks, ind), keep_logits=not fused)) for _ in range(1)]
        eng = m.unet.engine()
        plan = eng.last_plan
        assert (plan.head_ce_slots(10) > 0), "the head of this plan must be eligible for the fused launch"
        assert plan.logits_current is (not fused)
        torch.cuda.synchronize()
        out[fused] = (losses, plan.dlogits.t.clone(), plan._ctseg_loss.cnt.clone(), eng.store.flat_g.clone(), eng.store.flat_p.clone(),
                      float(m.logged["Mean Dice Score (train)"]))
    a, b = out[False], out[True]
    assert plan.dlogits.ld == (12 if narrow == "1" else 16)
    assert torch.equal(a[2], b[2]), "Dice counts"
    da, db = a[1].float(), b[1].float()
    assert float((da != db).float().mean()) < 0.02, "d loss / d logits: identical but for rounding"
    assert float((da - db).abs().max()) <= 2.0 ** -7 * float(da.abs().max())          # one bf16 ulp of the largest element
    np.testing.assert_allclose(b[0], a[0], rtol=2e-6)
    assert a[5] == b[5]
    ga, gb = a[3].double(), b[3].double()
    assert float(torch.dot(ga, gb) / (ga.norm() * gb.norm())) > 0.999999


def test_fused_head_at_full_size_matches_two_pass_loss_and_counts():
    """BASELINE.json's metric shape (2 x 512 x 512 x 48, bf16, narrow head): one fused step against one two-pass step from the
    same weights — loss to 2e-6, Dice counts bit-identical, the flat gradient equal in direction to 1e-6."""
    from bench import synthetic_batch
    from capstone_amd.volumetric.base_trainer import BaseUNet3D
    batch = synthetic_batch(2, 512, 512, 48, torch.device(DEV), 12342)
    out = {}
    for fused in (False, True):
        torch.manual_seed(12342)
        m = BaseUNet3D(filters=[32, 64, 128, 256], loss_fx=["CrossEntropy"], precision="bf16").to(DEV)
        l = float(m.fit_step(batch, keep_logits=not fused))
        torch.cuda.synchronize()
        out[fused] = (l, m.unet.engine().last_plan._ctseg_loss.cnt.clone(), m.unet.engine().store.flat_g.clone())
        del m
    assert abs(out[True][0] - out[False][0]) <= 2e-6 * abs(out[False][0])
    assert torch.equal(out[True][1], out[False][1])
    ga, gb = out[True][2].double(), out[False][2].double()
    assert float(torch.dot(ga, gb) / (ga.norm() * gb.norm())) > 0.999999


@pytest.mark.parametrize("nres", [0, 2])
def test_2d_downsample_conv1x1_trains_on_gpu(nres):
    """--downsample (capstone/training/base_trainer.py:53,81-85) on the GPU: the trainable 3 -> 1 channel mix in front of the 2-D
    U-Net gets its gradient from the stem's input-gradient pass (a 4-class stride-2 pass with ONE output column)."""
    from capstone_amd.training.base_trainer import BaseUNet2D
    from oracle import losses as OL, metrics as OM
    from oracle.monai_unet import UNet as OracleUNet
    torch.manual_seed(11)
    filters = [8, 16, 24, 32, 48]
    ref = OracleUNet(2, 1, 10, filters, (2, 2, 2, 2), num_res_units=nres)
    m = BaseUNet2D(filters=list(filters), use_res_units=nres > 0, downsample=True, loss_fx=["CrossEntropy", "Dice"], transform_degree=1)
    m.unet.load_state_dict(ref.state_dict())
    c1 = torch.nn.Conv2d(3, 1, 1)
    c1.load_state_dict(m.conv1x1.state_dict())
    m.to(DEV)
    g = torch.Generator().manual_seed(12)
    images = torch.randn(2, 3, 64, 96, generator=g)
    masks = (torch.rand(2, 9, 64, 96, generator=g) < 0.1).to(torch.uint8)
    ind = torch.ones(2, 9)
    labels = OM.squash_masks(masks, 10)
    rv = OL.MultipleLoss(["CrossEntropy", "Dice"])(ref(c1(images)), labels, ind)
    lref = torch.stack(list(rv.values())).sum()
    lref.backward()
    loss = m.training_step((images.to(DEV), masks.to(DEV), ind.to(DEV)))
    loss.backward()
    np.testing.assert_allclose(loss.item(), lref.item(), rtol=1e-4)
    gw, gb = m.conv1x1.weight.grad.cpu().numpy(), m.conv1x1.bias.grad.cpu().numpy()
    np.testing.assert_allclose(gw, c1.weight.grad.numpy(), rtol=2e-3, atol=1e-6)
    np.testing.assert_allclose(gb, c1.bias.grad.numpy(), rtol=2e-3, atol=1e-6)
    assert float(np.abs(c1.weight.grad.numpy()).max()) > 1e-4


@pytest.mark.parametrize("cout,shape", [(10, (2, 5, 6, 9)), (16, (1, 4, 4, 8)), (10, (1, 9, 13, 17)), (12, (3, 1, 2, 3))])
def test_transposed_conv_weight_gradient_lds_halo_kernel(monkeypatch, cout, shape):
    """conv_wgrad_up_kernel (ConvTranspose3d 64 -> <= 16, k3 s2; reference layer: MONAI UNet's top up-sampling block behind
    capstone/models/unet.py) vs torch on the CPU and vs the generic split-K kernel it replaces: ragged tiles on every axis,
    several samples, extents smaller than one tile."""
    from capstone_amd._native import BF16
    from helpers import run_conv_module, rel_err
    torch.manual_seed(cout + shape[1])
    mod = torch.nn.ConvTranspose3d(64, cout, 3, 2, 1, output_padding=1)
    x = torch.randn(shape[0], 64, *shape[1:])
    xr = x.clone().requires_grad_(True)
    y = mod(xr)
    gy = torch.randn_like(y)
    y.backward(gy)
    _, _, gw, _ = run_conv_module(mod, x, gy, BF16, DEV)
    assert rel_err(gw, mod.weight.grad) < 2.5e-2
    monkeypatch.setenv("CTSEG_NO_WGRAD_UP", "1")
    _, _, gw_generic, _ = run_conv_module(mod, x, gy, BF16, DEV)
    # same bf16 operands, fp32 accumulation in another order
    assert rel_err(gw, gw_generic) < 2e-5, rel_err(gw, gw_generic)


# ----------------------------------------------------------------------------------------------------------------------
# InstanceNorm + PReLU of the head's transposed conv applied by its consumers on load (ctseg_conv_desc::in_mean_rstd)
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape,fused_ce", [((2, 32, 48, 16), True), ((1, 36, 44, 20), False), ((3, 20, 24, 12), True)])
def test_head_norm_on_load_is_bit_identical_to_the_materialised_activation(monkeypatch, shape, fused_ce):
    """The activation between the top ConvTranspose3d (+ InstanceNorm + PReLU; reference: MONAI UNet up block behind
    capstone/models/unet.py) and the logits residual unit is not written: the logits convolution (forward, fused with the
    cross-entropy or not), its identity residual and its weight gradient normalise the raw conv output while staging it.  Same
    arithmetic, same rounding to bf16 => the loss, d loss / d logits, every gradient and the updated parameters are BIT-identical
    to the plan that records the apply pass (CTSEG_NORM_ON_LOAD=0).  Ragged tiles, several samples."""
    from capstone_amd.volumetric.base_trainer import BaseUNet3D
    B, H, W, D = shape
    g = torch.Generator().manual_seed(77)
    images = torch.randn(B, 1, H, W, D, generator=g).to(DEV)
    masks = (torch.rand(B, 9, H, W, D, generator=g) < 0.08).to(torch.uint8).to(DEV)
    ind = torch.ones(B, 9, dtype=torch.float64).to(DEV)
    out = {}
    for on_load in ("0", "1"):
        monkeypatch.setenv("CTSEG_NORM_ON_LOAD", on_load)
        torch.manual_seed(9)
        m = BaseUNet3D(filters=[16, 32, 64], loss_fx=["CrossEntropy"], precision="bf16").to(DEV)
        losses = [float(m.fit_step((images, masks, ind), keep_logits=not fused_ce)) for _ in range(2)]
        eng = m.unet.engine()
        plan = eng.last_plan
        names = [name for prog in (plan.fwd, plan.bwd) for name, *_ in prog]
        n_apply = sum(1 for nm in names if nm == "ctseg_instnorm_prelu_fwd")
        torch.cuda.synchronize()
        out[on_load] = (losses, plan.dlogits.t.clone(), eng.store.flat_g.clone(), eng.store.flat_p.clone(), n_apply)
    a, b = out["0"], out["1"]
    assert b[4] == a[4] - 1, "exactly one apply pass fewer"
    assert a[0] == b[0], "loss"
    assert torch.equal(a[1], b[1]), "d loss / d logits"
    assert torch.equal(a[2], b[2]), "gradients"
    assert torch.equal(a[3], b[3]), "parameters after two steps"


def test_head_norm_on_load_fp16_inference(monkeypatch):
    """the same fusion on the forward-only IEEE-half plan (configs[4] precision): logits bit-identical to the plan with the apply pass"""
    from capstone_amd.volumetric.base_trainer import BaseUNet3D
    g = torch.Generator().manual_seed(78)
    x = torch.randn(2, 1, 40, 32, 24, generator=g).to(DEV)
    outs = []
    for on_load in ("0", "1"):
        monkeypatch.setenv("CTSEG_NORM_ON_LOAD", on_load)
        torch.manual_seed(10)
        m = BaseUNet3D(filters=[16, 32, 64], loss_fx=["CrossEntropy"], precision="fp16").to(DEV).eval()
        with torch.no_grad():
            outs.append(m(x).float().clone())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("cout,shape", [(10, (2, 10, 16, 16)), (12, (1, 13, 13, 17)), (10, (1, 16, 16, 16))])
def test_head_transposed_conv_backward_reads_12_wide_gradient_rows(cout, shape):
    """dOut of the head's ConvTranspose3d(64 -> <= 12) laid out 12 elements wide (24-byte rows): its weight gradient
    (conv_wgrad_up_kernel<12>: 12-byte LDS-DMA pieces) and its input gradient (the stride-2 halo pass staged in 8-byte pieces)
    give the BITS of the 16-wide layout, and agree with torch on the CPU.  Ragged tiles, several samples."""
    from capstone_amd import _native as nat
    from capstone_amd.engine import GemmLayer
    from helpers import MiniPlan, to_cl, from_cl, rel_err
    torch.manual_seed(cout + shape[2])
    mod = torch.nn.ConvTranspose3d(64, cout, 3, 2, 1, output_padding=1)
    x = torch.randn(shape[0], 64, *shape[1:])
    xr = x.clone().requires_grad_(True)
    y = mod(xr)
    gy = torch.randn_like(y)
    y.backward(gy)
    ref_gx, ref_gw = xr.grad.clone(), mod.weight.grad.detach().clone()      # (the plans below re-home the parameters and their .grad)
    res = {}
    for ld in (16, 12):
        plan = MiniPlan([mod.weight, mod.bias], DEV, nat.BF16, 3)
        layer = GemmLayer(plan, "t", True, 3, 2, 64, [(mod.weight, mod.bias, cout)], 64)
        plan.packer.finalize()
        xa = to_cl(x, nat.BF16, DEV)
        layer.emit_fwd(xa)
        plan.run()
        ga = to_cl(gy, nat.BF16, DEV, ld=ld)
        gxa = layer.emit_dgrad(ga)
        plan.run()
        layer.emit_wgrad(xa, ga, bias_done=True)      # (the plan takes the bias gradient from the norm-backward pass)
        plan.run()
        torch.cuda.synchronize()
        res[ld] = (from_cl(gxa).clone(), plan.store.grad_view(mod.weight).cpu().clone())
    assert rel_err(res[12][0], ref_gx) < 2.5e-2 and rel_err(res[12][1], ref_gw) < 2.5e-2
    assert torch.equal(res[12][0], res[16][0]), "input gradient"
    assert torch.equal(res[12][1], res[16][1]), "weight gradient"


@pytest.mark.parametrize("shape", [(2, 32, 32, 16), (1, 18, 40, 24), (1, 64, 64, 8), (1, 8, 24, 96)])
def test_stride2_conv_32_to_128_with_register_resident_weights(monkeypatch, shape):
    """conv_down_r_kernel (Conv3d 32 -> 128 k3 s2: the fused [residual | unit0] convolution of the second down block, and the input
    gradient of the level-1 transposed conv; reference layers: MONAI UNet behind capstone/models/unet.py): forward, input gradient
    and weight gradient vs torch on the CPU, forward also vs the generic kernel it replaces (same bf16 operands, another sum order).
    Ragged 8 x 8 faces, each volume axis as the 1-deep tile axis, two samples."""
    from capstone_amd._native import BF16
    from helpers import run_conv_module, rel_err
    torch.manual_seed(shape[1] + shape[3])
    mod = torch.nn.Conv3d(32, 128, 3, 2, 1)
    x = torch.randn(shape[0], 32, *shape[1:])
    xr = x.clone().requires_grad_(True)
    y = mod(xr)
    gy = torch.randn_like(y)
    y.backward(gy)
    ref_y, ref_gx, ref_gw = y.detach().clone(), xr.grad.clone(), mod.weight.grad.detach().clone()
    yy, gx, gw, _ = run_conv_module(mod, x, gy, BF16, DEV)
    assert rel_err(yy, ref_y) < 2.5e-2 and rel_err(gx, ref_gx) < 2.5e-2 and rel_err(gw, ref_gw) < 2.5e-2
    monkeypatch.setenv("CTSEG_NO_DOWN_R", "1")
    yg, _, _, _ = run_conv_module(mod, x, gy, BF16, DEV)
    assert rel_err(yy, yg) < 1e-2          # both round fp32 sums of the same products to bf16


# ----------------------------------------------------------------------------------------------------------------------
# Generic weight gradient: multiply-free 32-bit gather addressing + range-checked buffer loads vs the 64-bit address chain
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind,cin,cout,shape", [("conv", 64, 64, (2, 16, 16, 12)), ("conv", 128, 256, (1, 9, 7, 5)),
                                                 ("conv_s2", 32, 128, (2, 18, 22, 10)), ("conv_s2", 64, 256, (1, 12, 14, 10)),
                                                 ("convT", 128, 32, (1, 5, 9, 7)), ("conv", 256, 256, (3, 4, 6, 5))])
def test_generic_weight_gradient_offset_addressing_equals_the_64_bit_chain(monkeypatch, kind, cin, cout, shape):
    """conv_wgrad_kernel's bf16 loader keeps scaled row coordinates and a 32-bit byte offset per chunk, advanced by per-stage constants
    with two carry corrections, and loads through range-checked buffer loads (voxels outside the volume, rows past the split and
    columns past d_valid are out-of-range offsets).  CTSEG_WGRAD_ADDR64=1 selects the 64-bit chain samples >= 2 GiB use: the same
    rows meet the same MFMAs in the same order, so weight and bias gradients must be BIT-identical (weight gradient of
    torch.nn.Conv3d / ConvTranspose3d as in the reference's loss.backward(), capstone/volumetric/base_trainer.py:80-82).
    Odd extents (carries on every axis inside a 32-row stage), several samples, stride 2, the transposed conv's swapped roles."""
    from capstone_amd._native import BF16
    from helpers import run_conv_module, rel_err
    torch.manual_seed(cin + cout + shape[2])
    if kind == "convT":
        mod = torch.nn.ConvTranspose3d(cin, cout, 3, 2, 1, output_padding=1)
    else:
        mod = torch.nn.Conv3d(cin, cout, 3, 2 if kind == "conv_s2" else 1, 1)
    x = torch.randn(shape[0], cin, *shape[1:])
    xr = x.clone().requires_grad_(True)
    y = mod(xr)
    gy = torch.randn_like(y)
    y.backward(gy)
    _, _, gw, gb = run_conv_module(mod, x, gy, BF16, DEV)
    assert rel_err(gw, mod.weight.grad) < 2.5e-2
    assert rel_err(gb, mod.bias.grad) < 2.5e-2
    monkeypatch.setenv("CTSEG_WGRAD_ADDR64", "1")
    _, _, gw64, gb64 = run_conv_module(mod, x, gy, BF16, DEV)
    assert np.array_equal(np.asarray(gw), np.asarray(gw64))
    assert np.array_equal(np.asarray(gb), np.asarray(gb64))


@pytest.mark.parametrize("kind,cin,cout,shape", [("convT", 384, 64, (1, 5, 6, 4)), ("conv_s2", 64, 256, (2, 12, 14, 10))])
def test_parity_classes_longest_first_equals_index_order(monkeypatch, kind, cin, cout, shape):
    """the generic 8-class passes (ConvTranspose3d 384 -> 64 forward; input gradient of the stride-2 Conv3d 64 -> 256) launch their
    classes longest K loop first; a class's tiles are independent of the launch order, so outputs are bit-identical to index order
    (CTSEG_CLASS_ORDER_KEEP=1)"""
    from capstone_amd._native import BF16
    from helpers import run_conv_module
    torch.manual_seed(cin + shape[1])
    if kind == "convT":
        mod = torch.nn.ConvTranspose3d(cin, cout, 3, 2, 1, output_padding=1)
    else:
        mod = torch.nn.Conv3d(cin, cout, 3, 2, 1)
    x = torch.randn(shape[0], cin, *shape[1:])
    gy = torch.randn_like(mod(x))
    y1, gx1, _, _ = run_conv_module(mod, x, gy, BF16, DEV)
    monkeypatch.setenv("CTSEG_CLASS_ORDER_KEEP", "1")
    y0, gx0, _, _ = run_conv_module(mod, x, gy, BF16, DEV)
    assert np.array_equal(np.asarray(y1), np.asarray(y0))
    assert np.array_equal(np.asarray(gx1), np.asarray(gx0))
