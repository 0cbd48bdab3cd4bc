"""GPU: whole-network forward/backward against the oracle over a spread of shapes that route the layers through every kernel
family (generic tiles, LDS-halo, streamed-weight halo, stride-2 halo, 8-class passes, stem) with ragged tiles and odd batch
sizes — a dispatch/eligibility regression net on top of the per-kernel tests.  Fixed seeds; the oracle runs in seconds."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CASES = [
    # (filters, batch, spatial, precision)
    ((16, 32, 64), 1, (40, 24, 12), "bf16"),          # 64-ch bottom at 10x6x3: generic; level-0 halo kernels; ragged tiles
    ((32, 64, 128), 2, (32, 32, 16), "bf16"),         # 64->64 streamed-weight halo, 8-class 128->32, 32-ch halo
    ((16, 64, 128), 3, (16, 48, 16), "bf16"),         # 16->64 stride-2 halo (input gradient of the 64->16 transposed conv), N = 3
    ((8, 16, 32, 64), 1, (48, 16, 40), "fp32"),       # fp32 MFMA path, 4 levels, z = 40 -> tiles of 8 exactly, y = 16
    ((32, 64), 2, (20, 36, 28), "fp32"),              # two levels, sizes divisible by 2 only
    ((16, 32, 64, 128), 1, (32, 64, 48), "bf16"),     # 4 levels; 12-deep style z at level 2 (48/4 = 12): permuted tile axes
    ((32, 64, 128, 256), 2, (32, 32, 16), "bf16"),    # the benchmark's channels: ring-pipelined 192x256 / 192x128 tiles on ragged
                                                      # 32- and 256-row grids, 12-wide rows around the logits conv
    ((32, 64, 128), 4, (16, 32, 16), "bf16"),         # N = 4: weight-gradient slab counts rounded to multiples of 8 (XCD-grouped grid)
]


@pytest.mark.parametrize("filters,B,sp,precision", CASES)
def test_network_forward_backward_vs_oracle(filters, B, sp, precision):
    import oracle.trainer as OT
    from capstone_amd.volumetric.base_trainer import BaseUNet3D
    torch.manual_seed(hash((filters, B, sp)) % 1000)
    om = OT.OracleUNet3D(filters=tuple(filters), loss_fx=("CrossEntropy",))
    with torch.no_grad():      # distinct PReLU slopes, so a mixed-up alpha shows
        for i, p in enumerate(om.parameters()):
            if p.numel() == 1:
                p.fill_(0.1 + 0.02 * (i % 11))
    m = BaseUNet3D(filters=list(filters), loss_fx=["CrossEntropy"], precision=precision)
    m.load_state_dict(om.state_dict())
    m.to(DEV)
    g = torch.Generator().manual_seed(5)
    images = torch.randn(B, 1, *sp, generator=g)
    masks = (torch.rand(B, 9, *sp, generator=g) < 0.08).to(torch.uint8)
    ind = torch.ones(B, 9, dtype=torch.float64)
    batch = (images, masks, ind)
    ologits = om(images)
    oloss = om.training_step(batch)
    oloss.backward()
    loss = m.fit_step(tuple(t.to(DEV) for t in batch))
    eng = m.unet.engine()
    got = eng.logits_view().cpu()
    scale = float(ologits.detach().abs().max())
    tol = 2e-4 if precision == "fp32" else 6e-2
    assert float((got - ologits.detach()).abs().max()) < tol * scale, "logits"
    assert abs(loss.item() - oloss.item()) < (1e-4 if precision == "fp32" else 2e-2) * max(1.0, abs(oloss.item())), "loss"
    worst = 1.0
    for (k, p), q in zip(om.named_parameters(), m.parameters()):
        a, b = eng.store.grad_view(q).cpu().flatten().double(), p.grad.flatten().double()
        if b.norm() > 1e-3 * max(1.0, b.numel() ** 0.5 * 1e-3):
            cos = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))
            worst = min(worst, cos)
            assert cos > (0.9999 if precision == "fp32" else 0.97), (k, cos)
    assert worst <= 1.0 + 1e-6
