"""Child process of tests/test_lightning_branch.py::test_lightning_configure_ddp_two_ranks: one RANK of a two-process gloo group.
Installs the pytorch_lightning stand-in, imports the product, and does what Lightning 1.0's DDP accelerator does with a
LightningModule (``ddp_accelerator.ddp_train``): ``init_ddp_connection`` -> module on its device -> ``configure_optimizers`` ->
``model = model.configure_ddp(model, device_ids)`` -> per batch ``trainer.model(batch, batch_idx)`` -> ``loss.backward()`` ->
``optimizer.step()``.  The reference reaches this through Trainer flags only (capstone/volumetric/base_trainer.py:196,217).
argv: rank world port.  Prints one JSON line."""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
for p in (HERE, REPO, os.path.join(REPO, "ct-image-segmentation_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

rank, world, port = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=port)

import lightning_stub  # noqa: E402

pl = lightning_stub.install()

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from abi_emulator import Emulator, patch_native  # noqa: E402
from capstone_amd import _native as nat  # noqa: E402
from capstone_amd import distributed as cdist  # noqa: E402
from capstone_amd import plan as plan_mod  # noqa: E402
from capstone_amd.volumetric import base_trainer as T3  # noqa: E402

emu = Emulator()
patch_native(nat, emu)
plan_mod.Plan.run = staticmethod(lambda prog, stream, lo=0, hi=None: emu.run(prog[lo:hi]))

cdist.init_from_env("gloo")                       # Lightning: model.init_ddp_connection(global_rank, world_size)
assert issubclass(T3.BaseUNet3D, pl.LightningModule)
torch.manual_seed(500 + rank)                     # ranks start DIFFERENT: configure_ddp must make them rank 0's replica
m = T3.BaseUNet3D(filters=[4, 8, 16], loss_fx=["Focal", "Dice"], lr=1e-2, batch_size=1, transform_degree=0)
g = torch.Generator().manual_seed(900 + rank)     # every rank its own shard of the batch (DistributedSampler)
images = torch.randn(1, 1, 8, 8, 8, generator=g)
masks = (torch.rand(1, 9, 8, 8, 8, generator=g) < 0.12).to(torch.uint8)
batch = (images, masks, torch.ones(1, 9, dtype=torch.float64))

out = {"rank": rank}
try:
    torch.nn.parallel.DistributedDataParallel(m)  # what Lightning's STOCK configure_ddp would build
    out["stock_ddp"] = "accepted"
except nat.NativeError:
    out["stock_ddp"] = "NativeError"

opt = m.configure_optimizers()
model = m.configure_ddp(m, [])                    # the hook under test
assert isinstance(model, cdist.NativeDataParallel) and model.module is m and m.reducer is not None
m._results = []
m.train()
losses = []
for i in range(3):
    opt.zero_grad()
    loss = model(batch, i)                        # -> training_step
    loss.backward()
    opt.step()
    losses.append(float(loss.detach()))
m.eval()
with torch.no_grad():
    assert model(batch, 0) is None                # -> validation_step
names = [r[0] for r in m._results]
assert "Mean Dice Score (val)" in names and "Dice Loss (train)" in names and "Focal Loss (train)" in names
st = m.unet.engine().store
out["weights_sha"] = hashlib.sha256(st.flat_p.numpy().tobytes()).hexdigest()
out["moments_sha"] = hashlib.sha256(st.adam_m.numpy().tobytes() + st.adam_v.numpy().tobytes()).hexdigest()
out["losses"], out["step"] = losses, st.step
d = m.epoch_dice_across_ranks("train")
out["global_dice"] = float(d[0])
dist.destroy_process_group()
print("LIGHTNING_DDP_CHILD " + json.dumps(out))
