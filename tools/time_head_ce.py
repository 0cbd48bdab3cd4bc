"""Times the fused head launch (logits convolution + cross-entropy, ctseg_conv_logits_ce) of the BASELINE shape in isolation.
Used by tools/ablate_head_ce.sh; honours the CTSEG_* switches of the plan (e.g. CTSEG_NORM_ON_LOAD)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ct-image-segmentation_amd")):
    sys.path.insert(0, p)
from bench import FILTERS, SEED, synthetic_batch
from capstone_amd.volumetric.base_trainer import BaseUNet3D
dev = torch.device("cuda:0")
torch.manual_seed(SEED)
m = BaseUNet3D(filters=list(FILTERS), loss_fx=["CrossEntropy"], precision="bf16").to(dev)
batch = synthetic_batch(2, 512, 512, 48, dev, SEED)
for _ in range(int(os.environ.get("CTSEG_HEADCE_WARM", "2"))): m.fit_step(batch, keep_logits=False)
plan = m.unet.engine().last_plan
le = plan._ctseg_loss
slots = plan.head_ce_slots(10)
dl = plan.dlogits
def run():
    le.prepare_fused_ce(False)
    le.head_ce(plan._head_ce[2], slots, dl.ptr(), dl.ld, weighted=False)
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print("fused head (incl. tiny prepare + reduce launches): %.4f ms" % (e0.elapsed_time(e1) / 20))
