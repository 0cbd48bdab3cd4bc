"""CPU: the ``_Base = pl.LightningModule`` branch of capstone_amd.volumetric.base_trainer / training.base_trainer — what every
reference user hits (capstone/volumetric/base_trainer.py:21 ``class BaseUNet3D(pl.LightningModule)``) — executed with a minimal
stand-in for the absent ``pytorch_lightning`` (tests/lightning_stub.py: frame-inspecting ``save_hyperparameters``, ``log`` that is a
no-op outside a loop and takes scalars only, ``device`` tracked through ``.to()``).  The stand-in has to be registered before the
product module is imported, so the scenario runs in a child interpreter (tests/lightning_child.py) on the C-ABI emulator."""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def test_lightning_module_branch_runs_training_step_and_fit_step():
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    p = subprocess.run([sys.executable, os.path.join(HERE, "lightning_child.py")], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-3000:] + "\n" + p.stderr[-6000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("LIGHTNING_CHILD ")][-1]
    out = json.loads(line[len("LIGHTNING_CHILD "):])
    assert out["base_is_lightning"] is True
    assert out["hparams"]["filters"] == [4, 8, 16, 32] and out["hparams"]["exclude_missing"] is False
    assert out["fit_step_losses"][1] < out["fit_step_losses"][0]
