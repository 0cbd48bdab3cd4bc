"""Shared test helpers: a one-layer plan around capstone_amd.engine.GemmLayer so single conv modules can be
driven through the C ABI exactly as the UNet plan drives them (same packing, taps, descriptors)."""
import numpy as np
import torch

from capstone_amd import _native as nat
from capstone_amd.engine import Act, GemmLayer, Packer, ParamStore, new_act, rup
from capstone_amd.plan import Plan, _NormAct


class MiniPlan:
    """duck-typed stand-in for plan.Plan holding one or more GemmLayers"""

    def __init__(self, params, device, dt, dims):
        self.device, self.dt, self.dims = torch.device(device), dt, dims
        self.store = ParamStore(params, self.device)
        self.packer = Packer(self)
        self.prog, self._keep, self.need_input_grad = [], [], True
        self.ready_marks = []

    def emit(self, name, *args, keep=()):
        self._keep.append((args, keep))
        self.prog.append((name, getattr(nat.lib(), name), tuple(args)))

    emit_colsum = Plan.emit_colsum

    def grads_ready(self, params):
        pass

    def run(self):
        self.packer.refresh(force=True)
        Plan.run(self.prog, nat.stream_ptr())
        self.prog = []


def to_cl(x, dt, device, ld=None):
    """(N,C,*sp) fp32 cpu tensor -> Act (channels-last storage on device)"""
    if x.ndim == 4:
        x = x.unsqueeze(-1)
    N, C = x.shape[:2]
    a = new_act(N, x.shape[2], x.shape[3], x.shape[4], C, dt, device, ld=ld)
    a.t[..., :C].copy_(x.permute(0, 2, 3, 4, 1).to(device))
    return a


def from_cl(a, two_d=False):
    v = a.valid().float().cpu()
    return v[..., 0] if two_d else v


def run_conv_module(mod, x, gy, dt, device):
    """forward, input-gradient and weight-gradient of one torch conv module through the HIP passes.
    Returns (y, gx, gw, gb) as fp32 cpu tensors in torch layout."""
    transposed = isinstance(mod, (torch.nn.ConvTranspose2d, torch.nn.ConvTranspose3d))
    dims = 3 if mod.weight.ndim == 5 else 2
    plan = MiniPlan([mod.weight, mod.bias], device, dt, dims)
    e = nat.epc(dt)
    cin = mod.in_channels
    xa = to_cl(x, dt, device, ld=cin if cin % e else None)
    layer = GemmLayer(plan, "t", transposed, mod.kernel_size[0], mod.stride[0], cin, [(mod.weight, mod.bias, mod.out_channels)],
                      cin if cin % e else rup(cin, e), need_dgrad=(cin % e == 0))
    plan.packer.finalize()
    y, _ = layer.emit_fwd(xa)
    plan.run()
    ga = to_cl(gy, dt, device)
    gx = None
    if layer.dg_pack is not None:
        gxa = layer.emit_dgrad(ga)
        plan.run()
        gx = from_cl(gxa, dims == 2)
    if dt == nat.F16:           # IEEE half storage: forward / input-gradient passes only (no weight-gradient kernels)
        torch.cuda.synchronize()
        return from_cl(y, dims == 2), gx, None, None
    layer.emit_wgrad(xa, ga)
    plan.run()
    torch.cuda.synchronize()
    gw = plan.store.grad_view(mod.weight).cpu().clone()
    gb = plan.store.grad_view(mod.bias).cpu().clone()
    return from_cl(y, dims == 2), gx, gw, gb


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))
