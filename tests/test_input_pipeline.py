"""3-D input pipeline: Resize3D + ToTensorV3 (+ fused _squash_masks_3D) on the device.

Golden vectors: tests/golden/pipeline3d.npz = outputs of the reference's own capstone/volumetric/transforms.py.
CPU: the oracle against them; the product's host logic through the numpy ABI emulator.  GPU: the HIP kernel, bit-exact."""
import os

import numpy as np
import pytest
import torch

from capstone_amd import _native as nat
from capstone_amd.volumetric import datasets as DS
from capstone_amd.volumetric import transforms as T
from capstone_amd.volumetric.utils import _squash_masks_3D
from oracle import input_pipeline as OP

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pipeline3d.npz")
CASES = ["up", "down", "mixed", "same"]


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(GOLD))


@pytest.mark.parametrize("tag", CASES)
def test_oracle_matches_reference_fixture(gold, tag):
    size = tuple(gold[f"{tag}_size"])
    img, m, lab = OP.instance(gold[f"{tag}_image"], gold[f"{tag}_masks"], size)
    assert np.array_equal(img.numpy(), gold[f"{tag}_image_out"])
    assert np.array_equal(m.numpy(), gold[f"{tag}_masks_out"])
    assert np.array_equal(lab.numpy(), gold[f"{tag}_labels"])
    D, H, W = gold[f"{tag}_image"].shape[1:]
    iD, iH, iW = (OP.nearest_index(o, i) for o, i in zip(size, (D, H, W)))      # the explicit index rule the kernel uses
    explicit = gold[f"{tag}_image"][0][iD][:, iH][:, :, iW].transpose(1, 2, 0)
    assert np.array_equal(explicit, gold[f"{tag}_image_out"][0])


def _check_product(gold, tag, dev):
    size = tuple(int(v) for v in gold[f"{tag}_size"])
    image = torch.from_numpy(gold[f"{tag}_image"]).to(dev)
    masks = torch.from_numpy(gold[f"{tag}_masks"]).to(dev)
    # reference composition: Compose([Resize3D(), ToTensorV3()]) applied per image / per mask
    rz, tt = T.Resize3D(size=size), T.ToTensorV3()
    mid = rz.apply(image)
    assert tuple(mid.shape) == (1,) + size
    img = tt.apply(mid)
    assert img.is_contiguous() and np.array_equal(img.cpu().numpy(), gold[f"{tag}_image_out"])
    m0 = tt.apply_to_mask(rz.apply_to_mask(masks[3]))
    assert np.array_equal(m0.cpu().numpy(), gold[f"{tag}_masks_out"][3])
    # fused instance pipeline, both forms
    out = T.InstancePipeline3D(size)(image=image, masks=list(masks))
    assert np.array_equal(out["image"].cpu().numpy(), gold[f"{tag}_image_out"])
    assert np.array_equal(out["masks"].cpu().numpy(), gold[f"{tag}_masks_out"])
    sq = T.InstancePipeline3D(size, squash=True)(image=image, masks=masks)
    assert np.array_equal(sq["masks"].cpu().numpy(), gold[f"{tag}_labels"].astype(np.uint8))
    assert np.array_equal(sq["hist"].cpu().numpy(), np.bincount(gold[f"{tag}_labels"].reshape(-1), minlength=10))
    # int16 raw volumes (NRRD CT) and the optional HU window
    i16 = torch.from_numpy(np.round(gold[f"{tag}_image"]).astype(np.int16)).to(dev)
    win = T.InstancePipeline3D(size, window=(400, 50))(image=i16, masks=masks)["image"].cpu().numpy()
    want = OP.instance(i16.cpu().numpy(), gold[f"{tag}_masks"], size, window=(400, 50))[0].numpy()
    assert np.array_equal(win, want)


@pytest.fixture()
def emu():
    from abi_emulator import Emulator, patch_native
    undo = patch_native(nat, Emulator())
    yield
    undo()


@pytest.mark.parametrize("tag", CASES)
def test_host_logic_emulated(emu, gold, tag):
    _check_product(gold, tag, "cpu")


def test_dataset_and_collate_emulated(emu, gold, tmp_path):
    d = tmp_path / "miccai_3d" / "train"
    d.mkdir(parents=True)
    for i, tag in enumerate(["up", "up"]):
        np.savez(d / f"p{i}.npz", image=gold[f"{tag}_image"] + i, masks=gold[f"{tag}_masks"].astype(bool if i else np.uint8),
                 mask_indicator=np.array([1, 1, 0, 1, 1, 1, 1, 1, 1.0]))
    size = tuple(int(v) for v in gold["up_size"])
    ds = DS.get_miccai_3d("train", T.InstancePipeline3D(size), root=str(tmp_path), device="cpu")
    assert len(ds) == 2
    images, masks, ind = DS.collate_3d([ds[0], ds[1]])
    assert images.shape == (2, 1, size[1], size[2], size[0]) and masks.shape == (2, 9, size[1], size[2], size[0])
    assert ind.shape == (2, 9) and ind.dtype == torch.float64
    assert np.array_equal(images[1].numpy(), gold["up_image_out"] + 1)
    lab_ref = _squash_masks_3D(masks, 10, "cpu")
    ds2 = DS.MiccaiDataset3D(str(d), T.InstancePipeline3D(size, squash=True), device="cpu")
    _, labels, _ = DS.collate_3d([ds2[0], ds2[1]])
    assert labels.dtype == torch.uint8 and labels.shape == (2, size[1], size[2], size[0])
    lab_fused = _squash_masks_3D(labels, 10, "cpu")            # recognised as pre-squashed: passes straight through
    assert torch.equal(lab_fused, lab_ref) and torch.equal(lab_fused[0], torch.from_numpy(gold["up_labels"]))
    assert torch.equal(lab_fused._ctseg_labels[1], lab_ref._ctseg_labels[1])


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("tag", CASES)
def test_gpu_kernel_bit_exact_vs_reference_fixture(gold, tag):
    _check_product(gold, tag, "cuda")


@pytest.mark.gpu
def test_gpu_full_size_instance_and_fused_training_batch():
    """a cropped-patient-sized instance (1x150x400x380) -> the reference's 96x256x256 target, bit-exact vs the oracle; then a
    training step fed by pre-squashed labels equals the step fed by the nine masks"""
    from capstone_amd.volumetric.base_trainer import BaseUNet3D
    g = torch.Generator().manual_seed(5)
    image = (torch.randn(1, 150, 400, 380, generator=g) * 400).to(torch.int16)
    masks = torch.zeros(9, 150, 400, 380, dtype=torch.uint8)
    for k in range(9):
        masks[k, 10 + 12 * k:40 + 12 * k, 50 + 30 * k:120 + 30 * k, 60 + 20 * k:200 + 20 * k] = 1
    size = (96, 256, 256)
    want_img, want_m, want_lab = OP.instance(image.float().numpy(), masks.numpy(), size)
    out = T.InstancePipeline3D(size)(image=image.cuda(), masks=masks.cuda())
    assert torch.equal(out["image"].cpu(), want_img) and torch.equal(out["masks"].cpu(), want_m)
    sq = T.InstancePipeline3D(size, squash=True)(image=image.cuda(), masks=masks.cuda())
    assert torch.equal(sq["masks"].cpu().long(), want_lab)
    assert torch.equal(sq["hist"].cpu(), torch.bincount(want_lab.reshape(-1), minlength=10))

    small = (16, 32, 32)
    ind = torch.ones(9, dtype=torch.float64).cuda()
    a = T.InstancePipeline3D(small)(image=image.cuda(), masks=masks.cuda())
    b = T.InstancePipeline3D(small, squash=True)(image=image.cuda(), masks=masks.cuda())
    b["masks"]._ctseg_hist = b["hist"]
    batch_a = DS.collate_3d([(a["image"], a["masks"], ind)] * 2)
    batch_b = DS.collate_3d([(b["image"], b["masks"], ind)] * 2)
    losses = []
    for batch in (batch_a, batch_b):
        torch.manual_seed(0)
        m = BaseUNet3D(filters=[8, 16, 32], loss_fx=["CrossEntropy", "Dice"]).cuda()
        losses.append(float(m.fit_step(batch)))
    assert losses[0] == losses[1]
