"""CPU: the oracle restatement against fixtures produced by the reference's own code."""
import numpy as np
import pytest
import torch

from oracle import losses as OL
from oracle import metrics as OM


@pytest.fixture(scope="module")
def leaf(golden):
    return golden("ref_leaf.npz")


def test_structures_and_weights(leaf):
    assert len(leaf["structures"]) == 9 and OM.N_CLASSES == 10
    np.testing.assert_array_equal(leaf["class_weight"], np.array(OL.CLASS_WEIGHT))


def test_squash_masks_bit_exact(leaf):
    got = OM.squash_masks(torch.from_numpy(leaf["squash_masks_in"]), 10)
    assert got.dtype == torch.int64
    np.testing.assert_array_equal(got.numpy(), leaf["squash_masks_out"])


def test_squash_predictions_ties(leaf):
    got = OM.squash_predictions(torch.from_numpy(leaf["squash_pred_in"]))
    np.testing.assert_array_equal(got.numpy(), leaf["squash_pred_out"])
    assert leaf["squash_pred_out"][0, 0, 0, 0] == 2 and leaf["squash_pred_out"][0, 1, 0, 0] == 5


def test_meandice_and_nan_path(leaf):
    p, t = torch.from_numpy(leaf["dice_pred"]), torch.from_numpy(leaf["dice_target"])
    score = OM.meandice(OM.one_hot(p.unsqueeze(1), 10), OM.one_hot(t.unsqueeze(1), 10), include_background=False)
    np.testing.assert_array_equal(score.numpy(), leaf["dice_score"])  # NaNs compare equal
    assert np.isnan(leaf["dice_score"]).any()


@pytest.mark.parametrize("mode", ["mean", "sum", "mean_batch", "sum_batch", "mean_channel", "sum_channel", "none"])
def test_metric_reduction_modes(leaf, mode):
    f, n = OM.metric_reduction(torch.from_numpy(leaf["dice_score"]).clone(), mode)
    np.testing.assert_array_equal(f.numpy(), leaf[f"reduce_{mode}_f"])
    np.testing.assert_array_equal(np.asarray(n), leaf[f"reduce_{mode}_n"])


def test_metric_reduction_rejects_unknown():
    with pytest.raises(ValueError):
        OM.metric_reduction(torch.zeros(2, 3), "median")


def test_dice_metric_wrapper(leaf):
    m, pc = OM.DiceMetric()(torch.from_numpy(leaf["dice_pred"]), torch.from_numpy(leaf["dice_target"]))
    np.testing.assert_array_equal(pc.numpy(), leaf["dice_per_class"])
    np.testing.assert_array_equal(m.numpy(), leaf["dice_mean"])


def test_dice_from_integer_counts_matches(leaf):
    p, t = leaf["dice_pred"], leaf["dice_target"]
    inter = np.stack([[((p[n] == c) & (t[n] == c)).sum() for c in range(1, 10)] for n in range(2)])
    pc = np.stack([[(p[n] == c).sum() for c in range(1, 10)] for n in range(2)])
    tc = np.stack([[(t[n] == c).sum() for c in range(1, 10)] for n in range(2)])
    m, per = OM.dice_from_counts(inter, pc, tc)
    np.testing.assert_array_equal(per.numpy(), leaf["dice_per_class"])
    np.testing.assert_array_equal(m.numpy(), leaf["dice_mean"])


def test_cross_entropy_values_and_grads(leaf):
    lg, tg = torch.from_numpy(leaf["ce_logits"]), torch.from_numpy(leaf["ce_target"])
    vals = OL.MultipleLoss(["CrossEntropy", "WeightedCrossEntropy"])(lg, tg, torch.ones(2, 9, dtype=torch.float64))
    np.testing.assert_array_equal(vals["CrossEntropy"].numpy(), leaf["ce_value"])
    np.testing.assert_array_equal(vals["WeightedCrossEntropy"].numpy(), leaf["wce_value"])
    for name, key in (("CrossEntropy", "ce_grad"), ("WeightedCrossEntropy", "wce_grad")):
        x = lg.clone().requires_grad_(True)
        OL.MultipleLoss([name])(x, tg)[name].backward()
        np.testing.assert_array_equal(x.grad.numpy(), leaf[key])


def test_generalized_dice_loss(leaf):
    lg, tg = torch.from_numpy(leaf["ce_logits"]), torch.from_numpy(leaf["gdl_target"])
    np.testing.assert_allclose(OL.generalized_dice_loss(lg, tg, "none").numpy(), leaf["gdl_none"], rtol=0, atol=0)
    np.testing.assert_allclose(OL.generalized_dice_loss(lg, tg, "mean").numpy(), leaf["gdl_mean"], rtol=0, atol=0)


def test_missing_mask_branches(leaf):
    t9, t10 = torch.from_numpy(leaf["mm_table"]), torch.from_numpy(leaf["mm_table10"])
    a, b = torch.from_numpy(leaf["mm_ind_a"]), torch.from_numpy(leaf["mm_ind_b"])
    np.testing.assert_array_equal(OL.missing_mask("Dice", t9, a).numpy(), leaf["mm_dice_a"])
    np.testing.assert_array_equal(OL.missing_mask("Dice", t9, b).numpy(), leaf["mm_dice_b"])
    np.testing.assert_array_equal(OL.missing_mask("Focal", t10, a).numpy(), leaf["mm_focal_a"])


def test_dice_and_focal_known_answers():
    """MONAI DiceLoss/FocalLoss are unpinned: check the restated formulas on hand-computable cases."""
    big = 40.0
    tg = torch.tensor([[[[1, 2], [0, 1]]]])  # labels (B=1, 1, 2, 2)
    lg = torch.full((1, 3, 1, 2, 2), -big)
    for idx, c in np.ndenumerate(tg[0].numpy()):
        lg[(0, int(c)) + idx] = big  # perfect, saturated prediction
    assert float(OL.dice_loss(lg, tg)) < 1e-5
    assert float(OL.focal_loss(lg, tg)) == 0
    wrong = lg.roll(1, dims=1)  # every voxel confidently wrong
    assert abs(float(OL.dice_loss(wrong, tg)) - 1.0) < 1e-4
    # one uniform voxel, 3 classes: p = 1/3 everywhere
    uni = torch.zeros(1, 3, 1, 1, 1)
    t1 = torch.tensor([[[[1]]]])
    # fg classes 1,2: inter=(1/3,0), denom=(1+1/3, 0+1/3)
    exp = torch.tensor([1 - (2 / 3 + 1e-5) / (4 / 3 + 1e-5), 1 - 1e-5 / (1 / 3 + 1e-5)])
    np.testing.assert_allclose(OL.dice_loss(uni, t1, "none")[0].numpy(), exp.numpy(), rtol=1e-6)
    fl = OL.focal_loss(uni, t1, "none")[0]
    np.testing.assert_allclose(fl.numpy(), [0, (2 / 3) ** 2 * np.log(3), 0], rtol=1e-6)
