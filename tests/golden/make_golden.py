#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ (run in the BUILD container only).

Nothing here travels to the GPU box except the .npz outputs. Three fixture families:

1. ``ref_leaf.npz``  — inputs/outputs of the reference's OWN code, executed from
   /root/reference by file path:
     * capstone/volumetric/utils.py::_squash_masks_3D and
       capstone/training/utils.py::_squash_predictions need only torch/numpy and are
       loaded unmodified, no stand-ins.
     * capstone/models/temp.py (compute_meandice, do_metric_reduction,
       GeneralizedDiceLoss), capstone/models/losses.py (CrossEntropyWrapper,
       WeightedCrossEntropyWrapper, MultipleLossWrapper, apply_missing_mask),
       capstone/models/metrics.py + capstone/volumetric/metrics.py
       (DiceMetricWrapper3D) import names from ``monai`` (absent in this image). They are
       loaded with an in-process namespace that supplies ONLY: the three string enums
       LossReduction/MetricReduction/Weight, ``one_hot`` (= scatter one-hot),
       ``AsDiscrete(to_onehot=True, n_classes)`` (= the same one-hot), inert
       DiceLoss/FocalLoss/UNet names that are never called, and
       ``capstone.utils.miccai.STRUCTURES`` read textually from the reference file.
       Every arithmetic statement executed is the reference's.
2. ``ops_torch.npz`` — op-level forward/backward of the torch CPU primitives MONAI composes
   (Conv3d k3 s1/s2, k1, ConvTranspose3d k3 s2 p1 op1, InstanceNorm3d, PReLU), odd sizes.
   ``pipeline3d.npz`` — the reference's 3-D input transforms (capstone/volumetric/transforms.py ``Resize3D`` +
   ``ToTensorV3``, the composition of volumetric/predefined.py:4-7) followed by ``_squash_masks_3D``; ``albumentations``
   is absent, so the file is loaded with a structural ``DualTransform`` base class (stores always_apply/p, nothing
   else): every statement executed in apply/apply_to_mask is the reference's.
3. ``unet_tiny.npz`` — end-to-end tiny 3-D U-Net step from the oracle restatement
   (weights stored in the fixture, so no RNG-order parity is needed on the GPU box).
   MONAI's UNet cannot be executed here => these vectors pin the build to the ORACLE,
   not to MONAI ("parity unpinned").
"""
import enum
import importlib.util
import os
import re
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
SEED = 12342
sys.path.insert(0, REPO)


def _load(name, relpath):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _install_namespace():
    def one_hot(labels, num_classes, dtype=torch.float, dim=1):
        shape = list(labels.shape)
        shape[dim] = num_classes
        return torch.zeros(shape, dtype=dtype, device=labels.device).scatter_(dim, labels.long(), 1)

    class AsDiscrete:
        def __init__(self, to_onehot=False, n_classes=None, **kw):
            self.to_onehot, self.n_classes = to_onehot, n_classes

        def __call__(self, x):
            return one_hot(x, self.n_classes) if self.to_onehot else x

    class LossReduction(enum.Enum):
        NONE, MEAN, SUM = "none", "mean", "sum"

    class MetricReduction(enum.Enum):
        NONE, MEAN, SUM = "none", "mean", "sum"
        MEAN_BATCH, SUM_BATCH = "mean_batch", "sum_batch"
        MEAN_CHANNEL, SUM_CHANNEL = "mean_channel", "sum_channel"

    class Weight(enum.Enum):
        SQUARE, SIMPLE, UNIFORM = "square", "simple", "uniform"

    class _Inert:  # never called by anything this script executes
        def __init__(self, *a, **k):
            raise RuntimeError("inert placeholder for an absent monai class was called")

    def mk(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mk("monai")
    mk("monai.networks", one_hot=one_hot)
    mk("monai.networks.nets", UNet=_Inert)
    mk("monai.utils", LossReduction=LossReduction, MetricReduction=MetricReduction, Weight=Weight)
    mk("monai.transforms", AsDiscrete=AsDiscrete)
    mk("monai.losses")
    mk("monai.losses.dice", DiceLoss=_Inert)
    mk("monai.losses.focal_loss", FocalLoss=_Inert)
    src = open(os.path.join(REF, "capstone/utils/miccai.py")).read()
    names = re.findall(r'"([A-Za-z_]+)"', re.search(r"STRUCTURES[^=]*=\s*\[(.*?)\]", src, re.S).group(1))
    assert len(names) == 9
    for pkg in ("capstone", "capstone.utils", "capstone.models", "capstone.volumetric"):
        mk(pkg).__path__ = []
    mk("capstone.utils.miccai", STRUCTURES=names)
    sys.modules["capstone.utils"].miccai = sys.modules["capstone.utils.miccai"]
    return names


def ref_leaf():
    g = torch.Generator().manual_seed(SEED)
    out = {}
    vu = _load("ref_vol_utils", "capstone/volumetric/utils.py")
    tu = _load("ref_train_utils", "capstone/training/utils.py")
    # --- squash masks: (2,9,6,5,4) uint8, overlapping on purpose (max => highest class wins)
    masks = (torch.rand(2, 9, 6, 5, 4, generator=g) < 0.15).to(torch.uint8)
    out["squash_masks_in"] = masks.numpy()
    out["squash_masks_out"] = vu._squash_masks_3D(masks, 10, "cpu").numpy()
    # --- squash predictions incl. engineered exact ties and sub-ulp near-ties
    logits = torch.randn(2, 10, 6, 5, 4, generator=g) * 3
    logits[0, 7, 0, 0, 0] = logits[0, 2, 0, 0, 0] = logits[0].max() + 1.0      # exact tie -> first index (2)
    logits[0, 5, 1, 0, 0] = 9.0
    logits[0, 8, 1, 0, 0] = 9.0                                                # exact tie -> 5
    logits[1, 3, 0, 0, 1] = 8.0
    logits[1, 6, 0, 0, 1] = float(np.nextafter(np.float32(8.0), np.float32(9.0)))  # 1 ulp above, later index
    out["squash_pred_in"] = logits.numpy()
    out["squash_pred_out"] = tu._squash_predictions(logits).numpy()

    names = _install_namespace()
    temp = _load("capstone.models.temp", "capstone/models/temp.py")
    losses = _load("capstone.models.losses", "capstone/models/losses.py")
    metrics = _load("capstone.models.metrics", "capstone/models/metrics.py")
    vmetrics = _load("capstone.volumetric.metrics", "capstone/volumetric/metrics.py")
    out["structures"] = np.array(names)
    out["class_weight"] = np.array(list(losses.WEIGHT.values()), dtype=np.float64)

    # --- Dice metric: class 4 absent from sample 0 (NaN path), class 9 absent everywhere
    target = torch.randint(0, 9, (2, 6, 5, 4), generator=g)
    target[0][target[0] == 4] = 0
    pred = torch.where(torch.rand(2, 6, 5, 4, generator=g) < 0.6, target, torch.randint(0, 10, (2, 6, 5, 4), generator=g))
    oh = sys.modules["monai.networks"].one_hot
    score = temp.compute_meandice(oh(pred.unsqueeze(1), 10), oh(target.unsqueeze(1), 10), include_background=False)
    out["dice_pred"], out["dice_target"], out["dice_score"] = pred.numpy(), target.numpy(), score.numpy()
    for mode in ("mean", "sum", "mean_batch", "sum_batch", "mean_channel", "sum_channel", "none"):
        f, nn_ = temp.do_metric_reduction(score.clone(), mode)
        out[f"reduce_{mode}_f"], out[f"reduce_{mode}_n"] = f.numpy(), np.asarray(nn_)
    dm, dpc = vmetrics.DiceMetricWrapper3D()(pred, target)
    out["dice_mean"], out["dice_per_class"] = dm.numpy(), dpc.numpy()

    # --- losses that the reference can run on 5-D input: CE, weighted CE (through MultipleLossWrapper)
    lg = torch.randn(2, 10, 8, 8, 4, generator=g) * 2
    tg = torch.randint(0, 10, (2, 8, 8, 4), generator=g)
    ind = torch.ones(2, 9, dtype=torch.float64)
    vals = losses.MultipleLossWrapper(["CrossEntropy", "WeightedCrossEntropy"])(lg, tg, ind)
    out["ce_logits"], out["ce_target"] = lg.numpy(), tg.numpy()
    out["ce_value"], out["wce_value"] = vals["CrossEntropy"].numpy(), vals["WeightedCrossEntropy"].numpy()
    lgr = lg.clone().requires_grad_(True)
    losses.MultipleLossWrapper(["CrossEntropy"])(lgr, tg)["CrossEntropy"].backward()
    out["ce_grad"] = lgr.grad.numpy()
    lgr = lg.clone().requires_grad_(True)
    losses.MultipleLossWrapper(["WeightedCrossEntropy"])(lgr, tg)["WeightedCrossEntropy"].backward()
    out["wce_grad"] = lgr.grad.numpy()

    # --- generalised Dice loss (reference-local copy, temp.py:17-170), 'none' and 'mean'
    tg2 = tg.clone()
    tg2[0][tg2[0] == 3] = 0  # empty class -> infinite weight branch (temp.py:150-153)
    for red in ("none", "mean"):
        gdl = temp.GeneralizedDiceLoss(include_background=False, to_onehot_y=True, softmax=True, reduction=red)
        out[f"gdl_{red}"] = gdl(lg, tg2.unsqueeze(1)).numpy()
    out["gdl_target"] = tg2.numpy()

    # --- apply_missing_mask: normal, the isinf branch, and the Focal (background-column) form
    table = torch.rand(3, 9, generator=g)
    ind_a = (torch.rand(3, 9, generator=g) < 0.8).float()
    ind_a[:, 0] = 1.0
    ind_a[0] = 1.0
    for c in range(9):
        if ind_a[:, c].sum() == 0:
            ind_a[0, c] = 1.0
    ind_b = ind_a.clone()
    ind_b[:, 2] = 0.0  # a class nobody annotated -> 1/0 = inf -> uniform weights
    table10 = torch.rand(3, 10, generator=g)
    out["mm_table"], out["mm_table10"] = table.numpy(), table10.numpy()
    out["mm_ind_a"], out["mm_ind_b"] = ind_a.numpy(), ind_b.numpy()
    out["mm_dice_a"] = losses.apply_missing_mask("Dice", table, ind_a).numpy()
    out["mm_dice_b"] = losses.apply_missing_mask("Dice", table, ind_b).numpy()
    out["mm_focal_a"] = losses.apply_missing_mask("Focal", table10, ind_a).numpy()
    np.savez_compressed(os.path.join(HERE, "ref_leaf.npz"), **out)
    print("ref_leaf.npz:", len(out), "arrays")


def pipeline3d():
    class DualTransform:      # structural stand-in for albumentations' base class: no arithmetic
        def __init__(self, always_apply=False, p=0.5):
            self.always_apply, self.p = always_apply, p

    for name in ("albumentations", "albumentations.core"):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    ti = types.ModuleType("albumentations.core.transforms_interface")
    ti.DualTransform = DualTransform
    sys.modules[ti.__name__] = ti
    tr = _load("ref_vol_transforms", "capstone/volumetric/transforms.py")
    vu = _load("ref_vol_utils2", "capstone/volumetric/utils.py")
    g = torch.Generator().manual_seed(SEED + 7)
    out = {}
    cases = {"up": ((5, 12, 10), (8, 16, 24)), "down": ((23, 37, 41), (8, 16, 16)), "mixed": ((9, 20, 33), (16, 8, 40)),
             "same": ((8, 16, 8), (8, 16, 8))}
    for tag, ((D, H, W), size) in cases.items():
        image = (torch.randn(1, D, H, W, generator=g) * 300).numpy().astype(np.float32)
        masks = (torch.rand(9, D, H, W, generator=g) < 0.2).to(torch.uint8).numpy()
        rz, tt = tr.Resize3D(size=size), tr.ToTensorV3()
        img_out = tt.apply(rz.apply(image))                                    # (1,H',W',D')
        m_out = torch.stack([tt.apply_to_mask(rz.apply_to_mask(m)) for m in masks])   # (9,H',W',D')
        lab = vu._squash_masks_3D(m_out.unsqueeze(0), 10, "cpu")[0]
        out[f"{tag}_size"] = np.array(size)
        out[f"{tag}_image"], out[f"{tag}_masks"] = image, masks
        out[f"{tag}_image_out"], out[f"{tag}_masks_out"], out[f"{tag}_labels"] = img_out.numpy(), m_out.numpy(), lab.numpy()
    np.savez_compressed(os.path.join(HERE, "pipeline3d.npz"), **out)
    print("pipeline3d.npz:", len(out), "arrays")


def ops_torch():
    g = torch.Generator().manual_seed(SEED + 1)
    out = {}

    def run(tag, mod, x):
        x = x.clone().requires_grad_(True)
        y = mod(x)
        gy = torch.randn(y.shape, generator=g)
        y.backward(gy)
        out[f"{tag}_x"], out[f"{tag}_y"], out[f"{tag}_gy"], out[f"{tag}_gx"] = (
            x.detach().numpy(), y.detach().numpy(), gy.numpy(), x.grad.numpy())
        for n, p in mod.named_parameters():
            out[f"{tag}_{n}"], out[f"{tag}_g{n}"] = p.detach().numpy(), p.grad.numpy()

    torch.manual_seed(SEED + 1)
    run("conv_k3s1", torch.nn.Conv3d(8, 12, 3, 1, 1), torch.randn(2, 8, 7, 6, 5, generator=g))
    run("conv_k3s2", torch.nn.Conv3d(8, 16, 3, 2, 1), torch.randn(2, 8, 10, 8, 6, generator=g))
    run("conv_k3s2_c1", torch.nn.Conv3d(1, 8, 3, 2, 1), torch.randn(1, 1, 12, 10, 8, generator=g))
    run("conv_k1", torch.nn.Conv3d(16, 24, 1, 1, 0), torch.randn(2, 16, 5, 4, 3, generator=g))
    run("convT", torch.nn.ConvTranspose3d(24, 8, 3, 2, 1, output_padding=1), torch.randn(2, 24, 5, 4, 3, generator=g))
    run("convT_c10", torch.nn.ConvTranspose3d(16, 10, 3, 2, 1, output_padding=1), torch.randn(1, 16, 6, 5, 4, generator=g))
    run("conv2d_k3s2", torch.nn.Conv2d(8, 8, 3, 2, 1), torch.randn(2, 8, 10, 12, generator=g))
    net = torch.nn.Sequential(torch.nn.InstanceNorm3d(8), torch.nn.PReLU())
    run("in_prelu", net, torch.randn(2, 8, 7, 6, 5, generator=g) * 2 + 0.7)
    np.savez_compressed(os.path.join(HERE, "ops_torch.npz"), **out)
    print("ops_torch.npz:", len(out), "arrays")


def _synthetic_batch(g, b, h, w, d):
    images = torch.randn(b, 1, h, w, d, generator=g)
    masks = torch.zeros(b, 9, h, w, d, dtype=torch.uint8)
    for n in range(b):
        for c in range(9):
            x0 = int(torch.randint(0, h - 4, (1,), generator=g))
            y0 = int(torch.randint(0, w - 4, (1,), generator=g))
            masks[n, c, x0:x0 + 4, y0:y0 + 4, (c % max(1, d - 2)):(c % max(1, d - 2)) + 2] = 1
    return images, masks, torch.ones(b, 9, dtype=torch.float64)


def unet_tiny():
    from oracle.trainer import OracleUNet3D

    out = {}
    for tag, filters, shape, losses in (
        ("a", (4, 8, 16, 32), (1, 16, 16, 16), ("CrossEntropy",)),
        ("b", (8, 8, 16, 16, 32), (2, 32, 32, 16), ("CrossEntropy", "Dice")),
    ):
        torch.manual_seed(SEED)
        g = torch.Generator().manual_seed(SEED + 2)
        m = OracleUNet3D(filters=filters, loss_fx=losses)
        batch = _synthetic_batch(g, *shape)
        sd0 = {k: v.clone() for k, v in m.state_dict().items()}
        opt = m.configure_optimizers()
        opt.zero_grad()
        _, labels, _, logits, total = m.shared_step(batch, True)
        total.backward()
        grads = {k: p.grad.clone() for k, p in m.named_parameters()}
        opt.step()
        out[f"{tag}_images"], out[f"{tag}_masks"], out[f"{tag}_indicator"] = (t.numpy() for t in batch)
        out[f"{tag}_labels"], out[f"{tag}_logits"] = labels.numpy(), logits.detach().numpy()
        out[f"{tag}_loss"] = total.detach().numpy()
        out[f"{tag}_dice_mean"] = m.logged["Mean Dice Score (train)"].numpy()
        out[f"{tag}_dice_per_class"] = m.logged["Dice per class (train)"].numpy()
        out[f"{tag}_filters"], out[f"{tag}_losses"] = np.array(filters), np.array(sorted(losses))
        for k, v in sd0.items():
            out[f"{tag}_w:{k}"] = v.numpy()
        for k, v in grads.items():
            out[f"{tag}_g:{k}"] = v.numpy()
        for k, v in m.state_dict().items():
            out[f"{tag}_w1:{k}"] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "unet_tiny.npz"), **out)
    print("unet_tiny.npz:", len(out), "arrays")


if __name__ == "__main__":
    torch.set_num_threads(4)
    which = sys.argv[1:] or ["ref_leaf", "pipeline3d", "ops_torch", "unet_tiny"]
    for w in which:
        globals()[w]()
