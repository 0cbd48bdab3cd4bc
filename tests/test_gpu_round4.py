"""GPU, round 4: the metric configuration as a TRAJECTORY — north_star's "mean Dice within +-0.002 of the CPU reference on fixed seeds"
at BASELINE.json's shape, over optimizer steps, in the dtype and through the step bench.py times."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def _dump(name, obj):
    try:
        os.makedirs(OUT, exist_ok=True)
        with open(os.path.join(OUT, name), "w") as f:
            json.dump(obj, f, indent=1)
    except OSError:
        pass


def test_full_size_bf16_fused_head_trajectory_tracks_the_oracle():
    """FIVE optimizer steps of the reference's 3-D recipe (capstone/volumetric/base_trainer.py:80-114: CrossEntropy, Adam lr 1e-3) on
    one 512 x 512 x 48 volume from the bench's generator: the torch-CPU oracle in fp32 against ``fit_step(keep_logits=False)`` in bf16
    storage (logits convolution with the cross-entropy in its epilogue, one-launch Adam) — per-step loss and mean Dice.  The single-step
    full-size tests (test_gpu_round2.py) pin one step; this pins that the UPDATE (Adam on the flat buffer, re-packed operands) carries
    the parity forward at the metric shape.  ~1 minute of host time for the oracle."""
    from bench import synthetic_batch
    import oracle.trainer as OT
    from capstone_amd.volumetric.base_trainer import BaseUNet3D
    steps = 5
    torch.manual_seed(12342)
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    om = OT.OracleUNet3D(filters=(32, 64, 128, 256), loss_fx=("CrossEntropy",), lr=1e-3)
    sd = {k: v.clone() for k, v in om.state_dict().items()}
    batch = synthetic_batch(1, 512, 512, 48, "cpu", 12342)
    opt = om.configure_optimizers()
    olosses, odices = [], []
    for _ in range(steps):
        olosses.append(float(om.fit_step(batch, opt)))
        odices.append(float(om.logged["Mean Dice Score (train)"]))
    del om, opt
    res = {"steps": steps, "oracle_loss": olosses, "oracle_dice": odices}
    gb = tuple(t.to(DEV) for t in batch)
    for precision, tol_l in (("bf16", 5e-3), ("fp32", 5e-4)):
        m = BaseUNet3D(filters=[32, 64, 128, 256], loss_fx=["CrossEntropy"], precision=precision, lr=1e-3)
        m.load_state_dict(sd)
        m.to(DEV)
        losses, dices = [], []
        for _ in range(steps):
            losses.append(float(m.fit_step(gb, keep_logits=False)))
            dices.append(float(m.logged["Mean Dice Score (train)"]))
        dl = [abs(a - b) / abs(b) for a, b in zip(losses, olosses)]
        dd = [abs(a - b) for a, b in zip(dices, odices)]
        res[precision] = {"loss": losses, "dice": dices, "max_rel_loss_diff": max(dl), "max_abs_dice_diff": max(dd)}
        _dump("parity_full_size_trajectory_vs_oracle.json", res)
        assert olosses[-1] < olosses[0], "the oracle's loss must descend"
        assert max(dd) <= 0.002, (precision, dd)
        assert max(dl) <= tol_l, (precision, dl)
        del m
        torch.cuda.empty_cache()
