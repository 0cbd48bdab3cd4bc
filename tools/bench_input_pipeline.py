"""Timing of the device input pipeline (row f1): raw instance 1x150x512x512 int16 + 9 masks -> 512x512x48 training volume.
Prints ms per instance and the GB/s over algorithmic bytes (target-side reads of the sources + writes)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ct-image-segmentation_amd"))
from capstone_amd.volumetric import transforms as T  # noqa: E402

src, size = (150, 512, 512), (48, 512, 512)
image = (torch.randn(1, *src, device="cuda") * 400).to(torch.int16)
masks = (torch.rand(9, *src, device="cuda") < 0.05).to(torch.uint8)
res = {}
for name, pipe in (("masks9", T.InstancePipeline3D(size)), ("fused_squash", T.InstancePipeline3D(size, squash=True))):
    for _ in range(3):
        pipe(image=image, masks=masks)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        pipe(image=image, masks=masks)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    nvox = size[0] * size[1] * size[2]
    byt = nvox * (2 + 9 + 4 + (9 if name == "masks9" else 1))
    res[name] = {"ms_per_instance": ms, "GBps_algorithmic": byt / ms / 1e6}
print(json.dumps(res))
