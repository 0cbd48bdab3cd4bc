import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ct-image-segmentation_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from capstone_amd import _native as nat
from capstone_amd.engine import GemmLayer
from capstone_amd.plan import _NormAct
from helpers import MiniPlan, to_cl, from_cl, rel_err
DEV = "cuda:0"
for kind, cin, cout, shape in (("convT", 128, 32, (1, 32, 32, 8)), ("convT", 128, 32, (1, 16, 16, 8)), ("convT", 128, 32, (2, 32, 32, 12)), ("conv", 64, 64, (1, 32, 32, 8))):
    torch.manual_seed(1)
    mod = torch.nn.ConvTranspose3d(cin, cout, 3, 2, 1, output_padding=1) if kind == "convT" else torch.nn.Conv3d(cin, cout, 3, 1, 1)
    alpha = torch.nn.Parameter(torch.tensor([0.25]))
    x = torch.randn(shape[0], cin, *shape[1:])
    plan = MiniPlan([mod.weight, mod.bias, alpha], DEV, nat.BF16, 3)
    layer = GemmLayer(plan, "t", kind == "convT", 3, 2 if kind == "convT" else 1, cin, [(mod.weight, mod.bias, cout)], cin)
    plan.packer.finalize()
    xa = to_cl(x, nat.BF16, DEV)
    y0, _ = layer.emit_fwd(xa)
    plan.run()
    torch.cuda.synchronize()
    print("no-stats nan", int(torch.isnan(from_cl(y0)).sum()))
    y, stats = layer.emit_fwd(xa, want_stats=True)
    out = _NormAct(plan, alpha).emit_fwd(y, stats, 0, None, None)
    plan.run()
    torch.cuda.synchronize()
    nz = torch.isnan(from_cl(y)).nonzero()
    print("nan at (n,c,x,y,z):", nz[:12].tolist())
    ref = mod.cpu()(x).detach()
    yy = from_cl(y)
    print(kind, cin, cout, shape, "y err", rel_err(yy, ref), "nan in y", int(torch.isnan(yy).sum()), "partials finite", bool(torch.isfinite(stats.partials).all()),
          "out err", rel_err(from_cl(out), torch.nn.functional.prelu(torch.nn.functional.instance_norm(ref), alpha.detach().cpu())))
