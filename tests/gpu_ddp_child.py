"""Child process of tests/test_gpu_two_rank.py: one RANK of a two-process group whose ranks share cuda:0 (the pool's boxes have one card),
transport gloo (RCCL refuses two ranks on one device).  Everything else is the product's N > 1 path on the real HIP kernels: replicas
made identical by attach(), the flat-buffer exchange fired from inside the recorded backward, the 1/world factor in the Adam kernel or
in publish().  argv: rank world port mode.  Prints one JSON line."""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
for p in (REPO, os.path.join(REPO, "ct-image-segmentation_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

rank, world, port, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                  CTSEG_DIST_BACKEND="gloo", CTSEG_SINGLE_DEVICE="1")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from capstone_amd import distributed as cdist  # noqa: E402
from capstone_amd.volumetric.base_trainer import BaseUNet3D  # noqa: E402

cdist.init_from_env()                      # the process group BEFORE the first GPU call
dev = torch.device("cuda:0")
torch.manual_seed(700 + rank)              # ranks start different
fused = mode == "fit_step_ce"
m = BaseUNet3D(filters=[16, 32, 64], loss_fx=["CrossEntropy"] if fused else ["Dice", "Focal"], precision="bf16", lr=1e-3).to(dev)
g = torch.Generator().manual_seed(40 + rank)
images = torch.randn(2, 1, 32, 48, 16, generator=g).to(dev)
masks = (torch.rand(2, 9, 32, 48, 16, generator=g) < 0.1).to(torch.uint8).to(dev)
batch = (images, masks, torch.ones(2, 9, dtype=torch.float64).to(dev))
model = m.configure_ddp(m, [0])            # ensure + attach + pass-through wrapper (Lightning 1.0's hook, called by hand)
m.train()
losses = []
if fused:
    for _ in range(3):
        losses.append(float(m.fit_step(batch, keep_logits=False)))
else:
    opt = m.configure_optimizers()
    for i in range(3):
        opt.zero_grad()
        loss = model(batch, i)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
torch.cuda.synchronize()
st = m.unet.engine().store
red = m.reducer
d = m.epoch_dice_across_ranks("train")
out = {"rank": rank, "mode": mode, "losses": losses, "step": st.step, "chunks": len(red.points_for(m.unet.engine().last_plan)) + 1,
       "weights_sha": hashlib.sha256(st.flat_p.cpu().numpy().tobytes()).hexdigest(),
       "moments_sha": hashlib.sha256(st.adam_m.cpu().numpy().tobytes() + st.adam_v.cpu().numpy().tobytes()).hexdigest(),
       "finite": bool(torch.isfinite(st.flat_p).all()), "global_dice": None if d is None else float(d[0])}
dist.destroy_process_group()
print("GPU_DDP_CHILD " + json.dumps(out))
