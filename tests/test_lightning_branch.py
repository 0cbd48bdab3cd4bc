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


def test_lightning_configure_ddp_two_ranks():
    """The reference's only door into multi-GPU (``Trainer`` flags ``--gpus N --distributed_backend ddp``,
    capstone/volumetric/base_trainer.py:196,217): Lightning 1.0's DDP accelerator calls ``model.configure_ddp(model, device_ids)``.
    Two gloo ranks with DIFFERENT initial weights and different data run that hook and three training_step -> backward -> step
    iterations with the paper's Dice + Focal recipe: replicas must end bit-identical, and a stock DistributedDataParallel wrap
    must be refused loudly."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "lightning_ddp_child.py"), str(r), "2", str(port)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(2)]
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=600)
        assert p.returncode == 0, so[-3000:] + "\n" + se[-6000:]
        line = [ln for ln in so.splitlines() if ln.startswith("LIGHTNING_DDP_CHILD ")][-1]
        outs.append(json.loads(line[len("LIGHTNING_DDP_CHILD "):]))
    a, b = sorted(outs, key=lambda o: o["rank"])
    assert a["stock_ddp"] == b["stock_ddp"] == "NativeError"
    assert a["weights_sha"] == b["weights_sha"] and a["moments_sha"] == b["moments_sha"]
    assert a["step"] == b["step"] == 3
    assert a["losses"] != b["losses"]                      # different shards: rank-local losses differ, the update does not
    assert a["global_dice"] == b["global_dice"]
