import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ct-image-segmentation_amd")):
    sys.path.insert(0, p)
from bench import synthetic_batch
from capstone_amd.volumetric.base_trainer import BaseUNet3D
dev = torch.device("cuda:0")
for filters, shape in (([16, 32, 64], (1, 128, 128, 32)), ([32, 64, 128, 256], (1, 128, 128, 32)), ([16, 32, 64], (1, 64, 64, 16)), ([16, 32, 64], (2, 64, 96, 32))):
    for on in ("0", "1"):
        os.environ["CTSEG_NORM_ON_LOAD"] = on
        torch.manual_seed(3)
        m = BaseUNet3D(filters=list(filters), loss_fx=["CrossEntropy"], precision="bf16").to(dev)
        batch = synthetic_batch(shape[0], *shape[1:], dev, 5)
        loss = m.fit_step(batch)
        lg = m.unet.engine().logits_view()
        print(filters, shape, on, float(loss), int(torch.isnan(lg).sum()), float(lg[torch.isfinite(lg)].abs().max()))
