"""Plan = the recorded forward/backward programs of one UNet for one (device, dtype, input shape).

Mirrors MONAI 0.3's ``UNet`` execution node by node:
  _ConvBlock      <- monai Convolution        (conv|convT -> InstanceNorm -> PReLU)
  _ResUnit        <- monai ResidualUnit       (conv(x) + residual(x))
  _Level          <- Sequential(down, SkipConnection(sub), up)
Each node emits C-ABI calls into ``plan.fwd`` / ``plan.bwd`` at build time; nothing here runs per step
except ``Plan.run``.
"""
import math
import os
import sys

import torch
import torch.nn as nn

from . import _native as nat
from ._native import BF16, F32
from .engine import Act, GemmLayer, Packer, ParamStore, new_act, rup


def _is_seq(m):
    """a plain nn.Sequential (a level, or up = Sequential(convT block, ResidualUnit)) — not a Convolution block"""
    return isinstance(m, nn.Sequential) and not hasattr(m, "conv_only")


def _parts(conv):
    return (conv.weight, conv.bias, conv.out_channels)


def _addend(out, accumulate):
    """accumulate: False | True (add the current content of ``out``) | an Act (add that tensor, ``out`` may be fresh)"""
    if accumulate is True:
        return out
    return accumulate if accumulate else None


class _NormAct:
    """InstanceNorm + PReLU of one conv output: forward apply and the 3-kernel backward."""

    def __init__(self, plan, alpha):
        self.plan, self.alpha = plan, alpha

    def defer(self, y, stats):
        """statistics only: the apply pass is left to the consumers of ``y`` (operand normalisation on load,
        ctseg_conv_desc::in_mean_rstd); ``materialise`` records it for a consumer that cannot"""
        self.y = y
        self.mr = stats.emit_finalize(0, y.C)
        self._applied = None
        y.pending_norm = self
        return y

    def materialise(self, y):
        if self._applied is None:
            plan = self.plan
            out = new_act(*y.dims, y.C, plan.dt, plan.device)
            plan.emit("ctseg_instnorm_prelu_fwd", plan.dt, y.ptr(), y.ld, self.mr.data_ptr(), plan.store.p_ptr(self.alpha),
                      None, 0, out.ptr(), out.ld, y.dims[0], y.S, y.C, keep=(y, out))
            self._applied = out
        return self._applied

    def emit_fwd(self, y, stats, col0, res, out):
        plan = self.plan
        self.y = y
        self.mr = stats.emit_finalize(col0, y.C)
        if out is None:
            out = new_act(*y.dims, y.C, plan.dt, plan.device)
        plan.emit("ctseg_instnorm_prelu_fwd", plan.dt, y.ptr(), y.ld, self.mr.data_ptr(), plan.store.p_ptr(self.alpha),
                  res.ptr() if res is not None else None, res.ld if res is not None else 0, out.ptr(), out.ld,
                  y.dims[0], y.S, y.C, keep=(y, res, out))
        return out

    def emit_bwd(self, g, dy_out=None, g_copy=None, colsum_out=None, apply=True):
        """g = dL/d(activation). Returns dL/dy (raw conv output); alpha's gradient goes to the flat buffer.
        ``colsum_out``: device pointer that receives sum over voxels of dL/dy per channel (bias gradient of a transposed
        conv feeding this norm) from the same pass.  ``apply=False``: statistics and slope gradient only — the consumer forms dL/dy
        on load (ctseg_wgrad_desc::dyn_*); returns the finalised sums instead."""
        plan, y = self.plan, self.y
        N, S, C = y.dims[0], y.S, y.C
        # row ranges per sample: 512 rows each on the big levels; on the small ones enough ranges for ~1024 workgroups in all (the
        # 25 MB bottom-level pass ran 96 workgroups: 70 us, 28 us with 1024; not below 32 rows per range)
        P = max(1, min(1024, math.ceil(S / 512)))
        if nat.is16(plan.dt):      # (fp32 storage keeps its partition: the 40-step fp32 trajectory test pins a summation order)
            P = max(P, min(math.ceil(1024 / N), math.ceil(S / 32)))
        ld = rup(C, 4)
        sums = torch.zeros((N, C, 2), dtype=torch.float32, device=plan.device)
        if dy_out is None and apply:
            dy_out = new_act(*y.dims, C, plan.dt, plan.device)
        a_ptr = plan.store.p_ptr(self.alpha)
        mark = getattr(g, "bst", None)
        fused = mark is not None and mark[0] is self
        rec = getattr(plan, "norm_bwd", None)      # (tests: the statistics of every norm's backward, and where they came from)
        if rec is not None:
            rec.append((sums, fused))
        if fused:
            # the pass that wrote g took the three sums in its epilogue (ctseg_conv_desc::bst_*): no reduce pass, no second read of g
            _, part, P, ld = mark
        else:
            part = torch.zeros((N, P, 3, ld), dtype=torch.float32, device=plan.device)
            plan.emit("ctseg_instnorm_prelu_bwd_reduce", plan.dt, g.ptr(), g.ld, y.ptr(), y.ld, self.mr.data_ptr(), a_ptr,
                      part.data_ptr(), P, ld, N, S, C, keep=(g, part))
        da_part = torch.zeros(N * C + 1, dtype=torch.float64, device=plan.device)
        # the slope-gradient sum over da_part rides on the apply pass below (one of its workgroups; no launch, no atomics)
        plan.emit("ctseg_instnorm_prelu_bwd_finalize", part.data_ptr(), N, P, ld, C, float(S), da_part.data_ptr(), sums.data_ptr(),
                  None, keep=(sums, da_part))
        if not apply:
            plan.emit("ctseg_instnorm_prelu_dalpha", da_part.data_ptr(), N * C, plan.store.g_ptr(self.alpha), keep=(da_part,))
            return sums
        args = (plan.dt, g.ptr(), g.ld, y.ptr(), y.ld, self.mr.data_ptr(), a_ptr, sums.data_ptr(), dy_out.ptr(), dy_out.ld,
                g_copy.ptr() if g_copy is not None else None, g_copy.ld if g_copy is not None else 0, N, S, C)
        slope = (da_part.data_ptr(), N * C, plan.store.g_ptr(self.alpha))
        if colsum_out is None:
            plan.emit("ctseg_instnorm_prelu_bwd_apply", *args, *slope, keep=(dy_out, g_copy))
        else:
            p_cap = N * 512
            cs_part = torch.zeros((p_cap, rup(C, nat.epc(plan.dt))), dtype=torch.float32, device=plan.device)
            plan.emit("ctseg_instnorm_prelu_bwd_apply_colsum", *args, cs_part.data_ptr(), p_cap, colsum_out, *slope,
                      keep=(dy_out, g_copy, cs_part))
        return dy_out


class _ConvBlock:
    """monai Convolution used as a whole layer (the up path's transposed conv; plain down layers when
    num_res_units == 0)."""

    def __init__(self, plan, mod, name, cg, need_dgrad=True):
        self.plan, self.mod = plan, mod
        self.gemm = GemmLayer(plan, name, mod.is_transposed, mod.kernel_size, mod.strides, mod.cin, [_parts(mod.conv)], cg,
                              need_dgrad)
        self.na = None if mod.conv_only else _NormAct(plan, mod.act.weight)
        self.params = [mod.conv.weight, mod.conv.bias] + ([] if mod.conv_only else [mod.act.weight])

    def emit_fwd(self, x, out=None, out_f32=False, defer_norm=False):
        self.x = x
        if self.na is None:
            y, _ = self.gemm.emit_fwd(x, out=out, out_f32=out_f32)
            return y
        y, stats = self.gemm.emit_fwd(x, want_stats=True)
        if defer_norm and out is None:
            return self.na.defer(y, stats)
        return self.na.emit_fwd(y, stats, 0, None, out)

    def first_bwd_norm(self):
        """the norm whose backward consumes the gradient handed to emit_bwd first (None: a convolution does)"""
        return self.na

    def emit_bwd(self, g, out=None, accumulate=False, need_dx=True, split_at=None, bst=None, bst_col0=0):
        """bst: the _NormAct that consumes the returned gradient (its channels from bst_col0 on) — its backward statistics are taken
        by the pass that writes that gradient where the kernel can (GemmLayer.emit_dgrad)"""
        bias = self.mod.conv.bias
        fuse_bias = self.na is not None and self.gemm.transposed and bias is not None
        if self.na is None:
            dy = g
        else:   # transposed conv: its bias gradient (sum of dOut over voxels) comes out of the norm's backward pass
            # dOut feeds the generic weight-gradient kernel and the stride-2 input-gradient pass: 16-byte chunked rows
            y = self.na.y
            # ... except where both consumers take 12-wide rows: the 64 -> <= 12 channel layer of the head (LDS-halo weight gradient
            # conv_wgrad_up_kernel<12> and the stride-2 halo pass staged in 8-byte pieces), a quarter fewer bytes in three passes
            gm = self.gemm
            narrow = (os.environ.get("CTSEG_NARROW_DOUT", "1") != "0" and self.plan.dt == BF16 and gm.transposed and gm.s == 2 and
                      gm.cin == 64 and self.x.ld == 64 and getattr(gm, "cgd", 0) == 16)
            dy_wide = new_act(*y.dims, y.C, self.plan.dt, self.plan.device, ld=None if narrow else rup(y.C, nat.epc(self.plan.dt)))
            dy = self.na.emit_bwd(g, dy_out=dy_wide, colsum_out=self.plan.store.g_ptr(bias) if fuse_bias else None)
        self.gemm.emit_wgrad(self.x, dy, bias_done=fuse_bias)
        self.plan.grads_ready(self.params)
        if not need_dx:
            return None
        return self.gemm.emit_dgrad(dy, out=out, add=_addend(out, accumulate), split_at=split_at, bst=bst, bst_col0=bst_col0)


class _ResUnit:
    def __init__(self, plan, mod, name, cg, need_dgrad=True):
        self.plan, self.mod = plan, mod
        units = list(mod.conv.children())
        self.units = units
        res = mod.residual
        self.fused = self.res_gemm = None
        self.identity = isinstance(res, nn.Identity)
        k, s = mod.kernel_size, mod.strides
        u0 = units[0]
        self.params = []
        if not self.identity and res.kernel_size[0] == k and not u0.conv_only:
            # residual conv and unit0 conv: same input, same geometry -> one GEMM with 2*C columns
            self.fused = GemmLayer(plan, name + ".res+unit0", False, k, s, mod.cin, [_parts(res), _parts(u0.conv)], cg, need_dgrad)
            self.gemms = [self.fused]
        else:
            if not self.identity:
                self.res_gemm = GemmLayer(plan, name + ".residual", False, res.kernel_size[0], s, mod.cin, [_parts(res)], cg,
                                          need_dgrad)
            self.gemms = [GemmLayer(plan, name + ".unit0", False, k, s, mod.cin, [_parts(u0.conv)], cg, need_dgrad)]
        e = nat.epc(plan.dt)
        for i, u in enumerate(units[1:], 1):
            self.gemms.append(GemmLayer(plan, f"{name}.unit{i}", False, k, 1, mod.cout, [_parts(u.conv)], rup(mod.cout, e)))
        self.nas = [None if u.conv_only else _NormAct(plan, u.act.weight) for u in units]

    def emit_fwd(self, x, out=None, out_f32=False):
        plan, C = self.plan, self.mod.cout
        self.x = x
        self.inputs, self.ys = [], []
        if self.identity:
            res = x
        elif self.fused is None:
            res, _ = self.res_gemm.emit_fwd(x)
        cur = x
        n = len(self.units)
        for i, (g, na) in enumerate(zip(self.gemms, self.nas)):
            last = i == n - 1
            self.inputs.append(cur)
            if na is None:  # last_conv_only: out = conv(cur) + bias + res, fused into the conv epilogue
                assert last and (self.fused is None or i > 0)
                y, _ = g.emit_fwd(cur, out=out, add=res, out_f32=out_f32)
                self.ys.append(y)
                return y
            yfull, stats = g.emit_fwd(cur, want_stats=True, split_at=C if (i == 0 and self.fused is not None) else None)
            if i == 0 and self.fused is not None:
                res, y, col0 = yfull.slice(0, C), yfull.slice(C, C), C
            else:
                y, col0 = yfull, 0
            self.ys.append(y)
            cur = na.emit_fwd(y, stats, col0, res if last else None, out if last else None)
        return cur

    def first_bwd_norm(self):
        return self.nas[-1]

    def emit_bwd(self, g, out=None, accumulate=False, need_dx=True, bst=None):
        """g = dL/d(out).  out = last_activation + res  =>  both receive g.  bst: see _ConvBlock.emit_bwd."""
        plan, C = self.plan, self.mod.cout
        n = len(self.units)
        d = g
        dfused = None
        # first layer of the network (no input gradient wanted): the weight-gradient pass of the fused [residual | unit0] convolution
        # reads d_res = g where it lies and forms d_y0 on load from unit0's norm — no apply pass for that norm, no copy of g
        dyn = (self.fused is not None and not need_dx and n >= 2 and self.nas[0] is not None and
               self.fused.wgrad_dyn_ok(self.x, g, self.ys[0]))
        if self.fused is not None and not dyn:
            dfused = new_act(*self.ys[0].dims, 2 * C, plan.dt, plan.device)   # [ d_res | d_y0 ]
        for i in range(n - 1, -1, -1):
            na, gm = self.nas[i], self.gemms[i]
            last = i == n - 1
            if na is None:
                dy = d
                if i == 0 and self.fused is not None:
                    raise NotImplementedError("conv_only unit fused with a strided residual")
            else:
                tgt = gcopy = None
                if i == 0 and dyn:
                    dyn_sums = na.emit_bwd(d, apply=False)
                    break
                if i == 0 and self.fused is not None:
                    tgt = dfused.slice(C, C)
                if last and self.fused is not None and not dyn:
                    gcopy = dfused.slice(0, C)      # hand g over to the fused dgrad/wgrad operand for free
                dy = na.emit_bwd(d, dy_out=tgt, g_copy=gcopy)
            if i == 0:
                break
            gm.emit_wgrad(self.inputs[i], dy)
            plan.grads_ready([gm.parts[0][0], gm.parts[0][1]] + ([na.alpha] if na is not None else []))
            d = gm.emit_dgrad(dy, bst=self.nas[i - 1])
        # ---- unit0 (+ residual branch) ----
        na0 = self.nas[0]
        alpha0 = [na0.alpha] if na0 is not None else []
        if self.fused is not None:
            if dyn:
                self.fused.emit_wgrad(self.x, g, dyn=(d, na0, dyn_sums))
            else:
                self.fused.emit_wgrad(self.x, dfused)
            plan.grads_ready([p for w, b, _ in self.fused.parts for p in (w, b)] + alpha0)
            if not need_dx:
                return None
            return self.fused.emit_dgrad(dfused, out=out, add=_addend(out, accumulate), bst=bst)
        g0 = self.gemms[0]
        g0.emit_wgrad(self.x, dy)
        ready = [g0.parts[0][0], g0.parts[0][1]] + alpha0
        if self.res_gemm is not None:
            self.res_gemm.emit_wgrad(self.x, g)
            ready += [self.res_gemm.parts[0][0], self.res_gemm.parts[0][1]]
        plan.grads_ready(ready)
        if not need_dx:
            return None
        if self.identity:
            if not accumulate:
                return g0.emit_dgrad(dy, out=out, add=g, bst=bst)        # dx = g + dgrad(dy)
            # identity residual AND an accumulated target (a bottom block with equal channel counts under a dense skip gradient):
            # dx = g + dgrad(dy) into a fresh tensor, then one elementwise pass adds the accumulated term
            tmp = g0.emit_dgrad(dy, add=g)
            acc = _addend(out, accumulate)
            if out is None:
                out = new_act(*tmp.dims, tmp.C, plan.dt, plan.device)
            plan.emit("ctseg_instnorm_prelu_fwd", plan.dt, tmp.ptr(), tmp.ld, None, None, acc.ptr(), acc.ld, out.ptr(), out.ld,
                      tmp.dims[0], tmp.S, tmp.C, keep=(tmp, acc, out))       # mean_rstd = NULL: out = tmp + acc
            return out
        if self.res_gemm is None:
            return g0.emit_dgrad(dy, out=out, add=_addend(out, accumulate), bst=bst)
        dx = g0.emit_dgrad(dy, out=out, add=_addend(out, accumulate))
        return self.res_gemm.emit_dgrad(g, out=dx, add=dx, bst=bst)      # += dgrad of the 1x1 residual conv (the pass that completes dx)


class _Level:
    """Sequential(down, SkipConnection(sub), up)"""

    def __init__(self, plan, seq, name, cg, is_top):
        self.plan, self.is_top = plan, is_top
        down, skip, up = seq[0], seq[1], seq[2]
        e = nat.epc(plan.dt)
        mk = lambda m, nm, c, nd=True: (_ResUnit if hasattr(m, "residual") else _ConvBlock)(plan, m, nm, c, nd)
        # construction order = MONAI's (sub, down, up); gradient-readiness order is up, sub, down
        sub = skip.submodule
        self.c1 = down.cout
        if _is_seq(sub):
            self.sub = _Level(plan, sub, name + ".1.submodule", rup(self.c1, e), False)
            self.c2 = sub[2][0].cout if _is_seq(sub[2]) else sub[2].cout
        else:
            self.sub = mk(sub, name + ".1.submodule", rup(self.c1, e))
            self.c2 = sub.cout
        self.down = mk(down, name + ".0", cg, not is_top or plan.need_input_grad)
        ccat = self.c1 + self.c2
        if _is_seq(up):
            self.up0 = _ConvBlock(plan, up[0], name + ".2.0", rup(ccat, e))
            self.up1 = _ResUnit(plan, up[1], name + ".2.1", rup(up[0].cout, e))
        else:
            self.up0, self.up1 = _ConvBlock(plan, up, name + ".2", rup(ccat, e)), None
        if nat.is16(plan.dt) and (self.c1 % e or self.c2 % e):
            raise nat.NativeError(f"16-bit storage needs channel counts that are multiples of {e} (got {self.c1}, {self.c2})")
        if plan.dt == F32 and (self.c1 % e or self.c2 % e):
            raise nat.NativeError(f"channel counts must be multiples of {e} (got {self.c1}, {self.c2})")

    def emit_fwd(self, x, out=None, out_f32=False):
        plan = self.plan
        od = self.down.gemms[0].out_dims(x.dims) if hasattr(self.down, "gemms") else self.down.gemm.out_dims(x.dims)
        self.cat = new_act(*od, self.c1 + self.c2, plan.dt, plan.device)
        xd = self.down.emit_fwd(x, out=self.cat.slice(0, self.c1))
        self.sub.emit_fwd(xd, out=self.cat.slice(self.c1, self.c2))
        if self.up1 is None:
            return self.up0.emit_fwd(self.cat, out=out, out_f32=out_f32)
        # the head (identity-residual unit whose only consumer of the activation is one 3x3x3 conv, its residual and its weight
        # gradient): the transposed conv's InstanceNorm + PReLU is applied by those passes on load, the activation is not written
        # MEASURED SLOWER at the reference's shape (round 2, 2 x 512 x 512 x 48 bf16): the apply pass it removes streams at HBM speed
        # (0.23 ms) while the ~30 VALU operations per staged 8-byte piece cost the fused logits conv + cross-entropy 0.80 -> 1.00 ms
        # and the logits weight gradient 0.42 -> 0.61 ms (10.49 -> 10.67 ms/step).  Off unless CTSEG_NORM_ON_LOAD=1.
        head = (os.environ.get("CTSEG_NORM_ON_LOAD", "0") == "1" and self.is_top and self.up1.identity and len(self.up1.units) == 1
                and self.up1.nas[0] is None and nat.is16(plan.dt))
        a = self.up0.emit_fwd(self.cat, defer_norm=head)
        return self.up1.emit_fwd(a, out=out, out_f32=out_f32)

    def first_bwd_norm(self):
        return self.up1.first_bwd_norm() if self.up1 is not None else self.up0.first_bwd_norm()

    def emit_bwd(self, g, out=None, accumulate=False, need_dx=True, depth=0, bst=None):
        plan = self.plan
        # The head's weight gradients (HBM-bound, 1.2 ms of side-stream work) are not queued beside the head's own HBM-bound
        # input-gradient / norm-backward passes but when the backward reaches level `defer_to`, whose passes are MFMA-bound:
        # same-box 12.48 -> 12.41 ms/step at 2, 12.44 at 1, 12.69 at 3 (the side stream's tail grows).
        # Round 2: with the head's weight gradients at 0.63 ms (x-column reuse, LDS-halo transposed-conv kernel) instead of 1.2 ms the
        # deferral no longer pays: not deferred 9.69 / 9.70 ms/step, level 1 9.90, level 2 9.79 / 9.83, level 3 10.00 (same box).
        defer_to = int(os.environ.get("CTSEG_DEFER_HEAD_WGRAD", "0"))   # 0 = off
        if self.is_top and defer_to > 0:
            plan._defer = []
        if depth == defer_to and depth > 0 and getattr(plan, "_stash", None):
            plan._defer, plan._stash = plan._stash, None
            plan.flush_deferred()
        if self.up1 is not None:
            g = self.up1.emit_bwd(g, bst=self.up0.first_bwd_norm())
        # [d(skip) | d(sub output)], as two dense tensors where the kernel can; the second half feeds the sub-block's last norm
        gcat = self.up0.emit_bwd(g, split_at=self.c1, bst=self.sub.first_bwd_norm(), bst_col0=self.c1)
        if self.is_top and defer_to > 0:
            plan._stash, plan._defer = plan._defer, None     # only the head's weight gradients wait
        # d(skip) = gcat[:, :c1] + d(sub input).  The sum goes to a DENSE tensor (the addend is read from the concat-gradient
        # slice): the norm-backward passes of the down block then stream full cache lines instead of half of every line
        if os.environ.get("CTSEG_DENSE_SKIP_GRAD", "1") != "0":
            kw = {"depth": depth + 1} if isinstance(self.sub, _Level) else {}
            gskip = self.sub.emit_bwd(gcat.slice(self.c1, self.c2), out=None, accumulate=gcat.slice(0, self.c1),
                                      bst=self.down.first_bwd_norm(), **kw)
        else:
            self.sub.emit_bwd(gcat.slice(self.c1, self.c2), out=gcat.slice(0, self.c1), accumulate=True)
            gskip = gcat.slice(0, self.c1)
        if self.is_top and getattr(plan, "_stash", None):     # never flushed below (fewer levels than asked for)
            plan._defer, plan._stash = plan._stash, None
            plan.flush_deferred()
        return self.down.emit_bwd(gskip, out=out, accumulate=accumulate, need_dx=need_dx, bst=bst)


def _make_side_stream(device):
    """the weight-gradient stream.  CTSEG_SIDE_PRIORITY=low|high: a HIP stream of the least / greatest priority the device offers
    (hipStreamCreateWithPriority through the runtime library torch already loaded), wrapped for torch; default: a plain stream."""
    want = os.environ.get("CTSEG_SIDE_PRIORITY", "")
    if want in ("low", "high"):
        # created by torch itself, i.e. in the HIP runtime torch has loaded (a second libamdhip64 bound by soname could be the
        # system's copy beside the wheel's: a stream handle of one runtime is invalid in the other).  torch: lower number = higher
        # priority; 0 is the default, negative values are clamped to the device's greatest priority.
        prio = 0 if want == "low" else -1
        if os.environ.get("CTSEG_SIDE_PRIORITY_VERBOSE"):
            print(f"side stream priority {prio} (torch convention: -1 = high, 0 = default / lowest)", file=sys.stderr)
        return torch.cuda.Stream(device=device, priority=prio)
    return torch.cuda.Stream(device=device)


class Plan:
    def __init__(self, engine, N, X, Y, Z, inference=False, need_input_grad=False, dt=None):
        net = engine.net
        self.inference = inference        # forward program only: no gradient buffers, no backward program
        self.engine, self.store, self.device = engine, engine.store, engine.device
        self.dt = engine.dt if dt is None else dt        # storage dtype of THIS plan (Engine.train_dt() for training plans)
        if self.dt == nat.F16 and not inference:
            # the weight-gradient / norm-backward / loss-gradient kernels exist for bf16 and fp32 storage only: without loss scaling
            # the gradients of this step underflow in half precision anyway (d loss / d logit ~ 1 / (B * voxels) = 4e-8 at
            # 2 x 512 x 512 x 48, below the smallest half subnormal 6e-8); bf16 has fp32's range at the same MFMA rate.
            # (Engine.plan_for_shape never asks for this: it records the training plans of an fp16 model in bf16, with a warning.)
            raise nat.NativeError("precision='fp16' (IEEE half storage) is implemented for inference: run under torch.no_grad() / "
                                  "the sliding-window inferer, and train with precision='bf16' or 'fp32'")
        self.dims = net.dimensions
        # gradient w.r.t. the input image: only the 2-D ``--downsample`` path needs it (a trainable 3->1 convolution sits in front
        # of the U-Net, capstone/training/base_trainer.py:53,81-85); the stem then gets an input-gradient operand
        self.need_input_grad = need_input_grad and not inference
        self.packer = Packer(self)
        self.fwd, self.bwd, self._cur = [], [], None
        self._defer = None
        self._keep = []
        self.ready_marks = []          # (program index in bwd, flat offset end) for gradient all-reduce overlap
        self.fwd_gen = 0               # forwards run on this plan's (single) set of activation buffers
        self.shape = (N, X, Y, Z)
        cin = net.in_channels
        self.root = _Level(self, net.model, "model", cin, True)
        # the pack of the first forward pass (the stem): rebuilt first after an optimizer step, see repack_after_update
        first = self.root.down.gemms[0] if hasattr(self.root.down, "gemms") else self.root.down.gemm
        self.packer.first = (first.fwd_pack["base"], first.fwd_pack["base"] + first.fwd_pack["size"])
        self.packer.finalize()
        self.norm_bwd = []
        # ---- record programs ----
        self.x = Act(torch.zeros((N, X, Y, Z, cin), dtype=nat.torch_dtype(self.dt), device=self.device), cin, 0, self.dt)
        self._cur = self.fwd
        self.logits = self.root.emit_fwd(self.x, out_f32=True)
        assert self.logits.t.dtype == torch.float32
        if not inference:
            self.dlogits = new_act(*self.logits.dims, net.out_channels, self.dt, self.device)
            self._cur = self.bwd
            self.dx = self.root.emit_bwd(self.dlogits, need_dx=self.need_input_grad)
            self._flush_reduces()
        self._cur = None

    # ---- recording ----
    def emit(self, name, *args, keep=()):
        fn = getattr(nat.lib(), name)
        conv = []
        for a in args:
            if isinstance(a, (nat.ConvDesc, nat.WgradDesc)):
                self._keep.append(a)
                conv.append(a)      # ctypes passes byref through the POINTER argtype
            else:
                conv.append(a)
        self._keep.append(keep)
        if name == "ctseg_conv_wgrad_reduce" and self._batch_reduce(conv):
            return
        op = (name, fn, tuple(conv))
        if self._defer is not None and self._cur is self.bwd and name in self.SIDE_OPS:
            self._defer.append(op)        # head weight gradients: queued later, beside the MFMA-bound deep levels
        else:
            self._cur.append(op)

    def emit_colsum(self, x, out_ptr):
        rows = x.dims[0] * x.S
        P = max(1, min(2048, math.ceil(rows / 2048)))
        part = torch.zeros((P, rup(x.C, 8)), dtype=torch.float32, device=self.device)
        self.emit("ctseg_colsum", self.dt, x.ptr(), x.ld, rows, x.C, part.data_ptr(), P, out_ptr, keep=(x, part))

    # ---- slab reduces of the weight gradients, batched ---------------------------------------------------------------------------
    # Every weight-gradient GEMM is followed by a reduce of its fp32 slabs into the flat gradient: 18 launches of 7-15 us per step
    # that occupied 0.44 ms of the weight-gradient stream (plus ~0.15 ms of gaps) in the overlapped trace of round 3 — each waits for
    # the tail of its GEMM and holds back the next one — on the stream that finishes the backward last.  They read nothing but their
    # own slabs and nothing reads their output before the gradient exchange / Adam, so they are collected and issued as ONE launch
    # (ctseg_conv_wgrad_reduce_batch, bit-identical sums) per gradient chunk: where the ready prefix of the flat gradient crosses the
    # split fractions of the data-parallel exchange (distributed.split_points: 60 %, 92 %) and at the end.  CTSEG_REDUCE_BATCH=0: off.
    def _batch_reduce(self, args):
        if (self._cur is not self.bwd or self._defer is not None or getattr(self, "_stash", None) or
                os.environ.get("CTSEG_REDUCE_BATCH", "1") == "0"):
            return False
        ws, nslabs, kpad_w, cn_pad, A, AS, T, col0, nb, dw, db = args
        if nat.lib().ctseg_conv_wgrad_reduce_batch_ok(ws, cn_pad, col0, nb) != 1:
            return False
        if not hasattr(self, "_pending_reduce"):
            self._pending_reduce, self._pending_ready, self._reduce_frac = [], [], 0
        self._pending_reduce.append(args)
        return True

    def _ready_prefix(self, extra):
        """elements of the flat gradient final once the offsets in `extra` are (plus everything marked ready so far)"""
        st = self.store
        done = {o for _, offs in self.ready_marks for o in offs} | set(extra)
        prefix = 0
        for p in st.params:
            if st.off(p) not in done:
                break
            prefix = st.off(p) + p.numel()
        return prefix

    def _flush_reduces(self, only_if_chunk=False):
        jobs = getattr(self, "_pending_reduce", None)
        if not jobs:
            return
        if only_if_chunk:
            from .distributed import SPLIT_FRACTIONS
            prefix, crossed = self._ready_prefix(self._pending_ready), False
            while self._reduce_frac < len(SPLIT_FRACTIONS) and prefix >= SPLIT_FRACTIONS[self._reduce_frac] * self.store.n:
                self._reduce_frac += 1
                crossed = True
            if not crossed:
                return
        arr = (nat.ReduceJob * len(jobs))()
        b0 = 0
        for J, (ws, nslabs, kpad_w, cn_pad, A, AS, T, col0, nb, dw, db) in zip(arr, jobs):
            J.ws, J.dw, J.db = ws, dw, db
            J.nslabs, J.kpad_w, J.cn_pad, J.A, J.Astride, J.T, J.col0, J.nb = nslabs, kpad_w, cn_pad, A, AS, T, col0, nb
            J.lanes = 8 if nslabs >= 64 else 32
            J.block0 = b0
            b0 += -(-((T * AS + 1) * ((nb + 3) // 4)) // J.lanes)
        table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone().to(self.device)
        ready, self._pending_reduce, self._pending_ready = self._pending_ready, [], []
        self.emit("ctseg_conv_wgrad_reduce_batch", table.data_ptr(), len(jobs), b0, keep=(table,))
        if ready:
            self.ready_marks.append((len(self.bwd), ready))

    def grads_ready(self, params):
        if self._cur is self.bwd:
            offs = [self.store.off(p) for p in params if p is not None]
            if self._defer is not None:
                self._defer.append(("ready", offs))
            elif getattr(self, "_pending_reduce", None):
                # (some of) these gradients come out of reduces that are still collected: final when the batch is issued
                self._pending_ready.extend(offs)
                self._flush_reduces(only_if_chunk=True)
            else:
                self.ready_marks.append((len(self.bwd), offs))

    def flush_deferred(self):
        """append the deferred side-stream ops (and their gradient-readiness marks) to the backward program"""
        ops, self._defer = self._defer, None
        for op in ops or ():
            if op[0] == "ready":
                self.ready_marks.append((len(self.bwd), op[1]))
            else:
                self.bwd.append(op)

    # ---- running ----
    @staticmethod
    def run(prog, stream, lo=0, hi=None):
        for name, fn, args in prog[lo:hi]:
            rc = fn(*args, stream)
            if rc != 0:
                nat.check(rc, name)

    def load_input(self, x):
        """(B,Cin,*sp) fp32 NC* tensor -> plan's channels-last storage buffer"""
        N, X, Y, Z = self.shape
        cin = self.x.C
        xs = x.reshape(N, cin, X * Y * Z)
        if not xs.is_contiguous() or xs.dtype != torch.float32:
            xs = xs.contiguous().float()
        nat.call("ctseg_nc_to_cl", xs.data_ptr(), self.x.t.data_ptr(), self.dt, N, cin, X * Y * Z, cin)

    def repack_after_update(self):
        """the optimizer has just rewritten the flat parameter buffer: rebuild the packed MFMA operands.  With a side stream
        the re-layout (0.07 ms, nothing else to overlap it with at the head of the next step) runs there, beside the loss
        bookkeeping and the next batch's input conversion; the next forward waits for its event."""
        side = self.side_stream()
        if side is None or os.environ.get("CTSEG_REPACK_SIDE", "1") == "0":
            self.packer.dirty = True
            return
        main = torch.cuda.current_stream(self.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            if self.packer.can_split() and os.environ.get("CTSEG_REPACK_SPLIT", "1") != "0":
                # the next step's first convolution needs its own operand and the biases only: everything else is rebuilt BESIDE
                # that convolution (the whole re-layout sat on the critical path between two steps: Adam -> repack -> stem)
                self.packer.refresh(force=True, part="first")
                self._repack_ev = side.record_event()
                self.packer.refresh(force=True, part="rest")
                self._repack_ev2 = side.record_event()
            else:
                self.packer.refresh(force=True)
                self._repack_ev = side.record_event()

    def head_ce_slots(self, n_classes):
        """partial-sum slots per sample of ctseg_conv_logits_ce for this plan's logits convolution (0: not eligible — the
        convolution and the loss then run as two passes)"""
        cache = getattr(self, "_head_ce", None)
        if cache is None or cache[0] != n_classes:
            slots = 0
            name, _, args = self.fwd[-1]
            if (name == "ctseg_conv_igemm" and not self.inference and args[0].out == self.logits.ptr() and
                    os.environ.get("CTSEG_FUSED_HEAD_CE", "1") != "0"):
                slots = nat.lib().ctseg_conv_logits_ce_slots(args[0], n_classes)
            cache = self._head_ce = (n_classes, max(slots, 0), args[0])
        return cache[1]

    def forward(self, x=None, skip_head=False):
        """x = None: the caller already filled ``self.x`` (sliding-window gather writes it directly).  skip_head: stop before the
        logits convolution (the caller runs it fused with the loss: ctseg_conv_logits_ce); ``self.logits`` is then NOT updated."""
        if x is not None:
            self.load_input(x)
        ev = getattr(self, "_repack_ev", None)
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
            self._repack_ev = None
        if self.packer.stale():
            # the parameters changed since repack_after_update queued its two-part re-layout (load_state_dict, a broadcast, a
            # foreign optimizer): refresh() is about to rewrite the packed buffer on THIS stream while the side stream's "rest" part
            # may still be reading the flat buffer and writing the same operands — order it behind that part first
            ev2 = getattr(self, "_repack_ev2", None)
            if ev2 is not None:
                torch.cuda.current_stream(self.device).wait_event(ev2)
                self._repack_ev2 = None
        self.packer.refresh()
        self.fwd_gen += 1
        # a d loss / d logits left by a fused loss whose backward never ran (skipped step, exception) belongs to the OLD batch
        self.dlogits_is_current = False
        end = len(self.fwd) - 1 if skip_head else None
        ev2 = getattr(self, "_repack_ev2", None)
        if ev2 is not None:
            self._repack_ev2 = None
            self.run(self.fwd, nat.stream_ptr(), 0, 1)           # the first pass (its operand is ready: _repack_ev)
            torch.cuda.current_stream(self.device).wait_event(ev2)
            self.run(self.fwd, nat.stream_ptr(), 1, end)
        else:
            self.run(self.fwd, nat.stream_ptr(), 0, end)
        self.logits_current = not skip_head
        return self.logits

    # weight-gradient work (split-K GEMM + slab reduce, transposed-conv bias sums) depends only on tensors that are final when
    # it is recorded and feeds nothing but the flat gradient buffer: it runs on a second HIP stream, so an LDS-port-bound
    # weight-gradient kernel shares the GPU with the HBM-bound norm passes / input-gradient convs of the next layers
    SIDE_OPS = ("ctseg_conv_wgrad", "ctseg_conv_wgrad_reduce", "ctseg_conv_wgrad_reduce_batch", "ctseg_colsum",
                "ctseg_instnorm_prelu_dalpha")

    def side_stream(self):
        """the second HIP stream of this plan (None on CPU / when CTSEG_SIDE_STREAM=0)"""
        if self.device.type != "cuda" or os.environ.get("CTSEG_SIDE_STREAM", "1") == "0" or self.inference:
            return None
        self._side_setup()
        return self._side

    def _side_setup(self):
        if getattr(self, "_side", None) is None:
            self._side = _make_side_stream(self.device)
            self._side_edges = [i for i in range(len(self.bwd) + 1)
                                if i == 0 or i == len(self.bwd) or
                                (self.bwd[i][0] in self.SIDE_OPS) != (self.bwd[i - 1][0] in self.SIDE_OPS)]
            self._side_ev = {}
            self._join_ev = torch.cuda.Event()
            # (Running the LAST group of weight-gradient ops — the first layer's — on the main stream, which is idle by then while the
            # side stream still works on the layer before: measured 9.05 -> 9.11 ms/step, three interleaved pairs.  Not done.)
            # (The 18 slab reduces on a THIRD stream, each behind an event of its GEMM, so that they stop sitting between two GEMMs of the
            # side stream — 0.69 ms of in-step side-stream time for launches that take 0.2 ms alone: 9.32 -> 9.52 ms/step, three pairs.)

    def backward(self, hooks=None, before_join=None):
        """runs the recorded backward; ``hooks`` = {program index: callable} fire between ops (DDP overlap).  ``before_join``:
        called once every op is queued but before the main stream waits for the side stream's tail (the last weight gradients,
        ~0.15 ms): main-stream work that does not need the gradients (scalar bookkeeping) goes there for free."""
        if self.inference:
            raise RuntimeError("this plan was recorded for inference (torch.no_grad()); run the forward with gradients enabled first")
        hooks = hooks or {}
        if self.device.type != "cuda" or os.environ.get("CTSEG_SIDE_STREAM", "1") == "0":
            st = nat.stream_ptr()
            lo = 0
            for idx in sorted(hooks):
                self.run(self.bwd, st, lo, idx)
                hooks[idx]()
                lo = idx
            self.run(self.bwd, st, lo)
            if before_join is not None:
                before_join()
            return
        self._side_setup()
        main, side = torch.cuda.current_stream(self.device), self._side
        st, ss = main.cuda_stream, side.cuda_stream
        busy = False

        def join():          # gradients written on the side stream become visible to what follows on the main stream
            nonlocal busy
            if busy:
                self._join_ev.record(side)
                main.wait_event(self._join_ev)
                busy = False

        n = len(self.bwd)
        bounds = sorted(set(self._side_edges) | {i for i in hooks if 0 <= i <= n})
        for a, b in zip(bounds[:-1], bounds[1:]):
            if a in hooks:
                # the hook (a gradient all-reduce) needs what BOTH streams have produced so far, but nothing on the main stream
                # needs the hook: it is issued from the side stream (which first waits for the main stream's position), so the
                # main stream never stalls on the weight gradients still queued there
                ev = self._side_ev.get(("hook", a))
                if ev is None:
                    ev = self._side_ev[("hook", a)] = torch.cuda.Event()
                ev.record(main)
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    hooks[a]()
                busy = True
            if self.bwd[a][0] in self.SIDE_OPS:
                if os.environ.get("CTSEG_TIMING_SKIP_SIDE") == "1":      # timing-only: what the weight-gradient stream costs the step
                    continue
                ev = self._side_ev.get(a)
                if ev is None:
                    ev = self._side_ev[a] = torch.cuda.Event()
                ev.record(main)                  # everything these ops read has been produced by now
                side.wait_event(ev)
                self.run(self.bwd, ss, a, b)
                busy = True
            else:
                self.run(self.bwd, st, a, b)
        if before_join is not None:
            before_join()
        join()
        if n in hooks:
            hooks[n]()


class Engine:
    """Owns the flat parameter store of one UNet and a cache of plans keyed by input shape."""

    def __init__(self, net):
        self.net = net
        self.dt = nat.DT_OF_PRECISION[net.precision]
        self.store = None
        self.device = None
        self.plans = {}
        self.last_plan = None
        self.reducer = None            # capstone_amd.distributed.GradAllReducer once the module is data-parallel (attach())
        self._warned_fp16_train = False
        self._warned_unattached = False

    def train_dt(self):
        """storage dtype of the TRAINING plans.  A model built with Lightning's ``--precision 16`` (IEEE half) runs its inference
        plans in fp16 and trains in bf16 storage — the reference's stack trains under that flag with native AMP + loss scaling
        (capstone/volumetric/base_trainer.py:217 exposes it); here there is no loss scaling and no fp16 backward kernel family,
        and bf16 has fp32's exponent range at the same MFMA rate.  Said once, loudly; never silent."""
        if self.dt != nat.F16:
            return self.dt
        if not self._warned_fp16_train:
            import warnings
            warnings.warn("precision=16 (IEEE half): training plans run in bf16 storage / fp32 accumulation (no loss scaling is "
                          "needed); validation, test and sliding-window inference run in fp16 storage as asked. Pass "
                          "precision='bf16' to use one storage dtype throughout.", RuntimeWarning, stacklevel=3)
            self._warned_fp16_train = True
        return BF16

    def warn_if_unattached(self):
        """a process group with several ranks exists but nobody made this model data-parallel: every rank then trains on its own
        gradients and the replicas drift apart without an error.  Said once per engine; ``CTSEG_LOCAL_GRADS=1`` silences it
        (rank-local warm-up steps, independent replicas)."""
        if self._warned_unattached or os.environ.get("CTSEG_LOCAL_GRADS") == "1":
            return
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            import warnings
            warnings.warn("torch.distributed runs with %d ranks but this model was not made data-parallel: gradients stay "
                          "rank-local. Call capstone_amd.distributed.attach(module) (Lightning: configure_ddp does it); a stock "
                          "DistributedDataParallel wrap cannot average gradients that the HIP kernels write into the flat buffer."
                          % dist.get_world_size(), RuntimeWarning, stacklevel=3)
            self._warned_unattached = True

    def _param_order(self):
        """backward (gradient-readiness) order: up path of the top level first, stem last"""
        order = []

        def level(seq):
            down, skip, up = seq[0], seq[1], seq[2]
            order.extend(reversed(list(up.parameters())))
            sub = skip.submodule
            if _is_seq(sub):
                level(sub)
            else:
                order.extend(reversed(list(sub.parameters())))
            order.extend(reversed(list(down.parameters())))

        level(self.net.model)
        assert len(order) == len(list(self.net.parameters()))
        return order

    def ensure(self, device):
        device = torch.device(device)
        if self.store is None or self.device != device:
            old = self.store
            self.device = device
            self.store = ParamStore(self._param_order(), device)
            self.plans = {}
            if self.reducer is not None:
                # a data-parallel module changed device: the exchange follows the new flat gradient buffer (split points are
                # computed per plan from its own readiness marks)
                from . import distributed as cdist
                red = self.reducer
                self.reducer = cdist.GradAllReducer(self.store.flat_g, self.store.n, None,
                                                    {self.store.off(p): p.numel() for p in self.store.params}, red.group, red.always)
            if old is not None and old.step > 0 and old.adam_m is not None and old.n == self.store.n:
                # the module changed device with optimizer state in the old store (a checkpoint loaded on the CPU, then .to(cuda);
                # Lightning calls on_load_checkpoint before it moves the module): Adam's moments and step count move with it.
                # The parameter VALUES come from the Parameters themselves (ParamStore.__init__), which .to() has moved.
                self.store.ensure_adam_state()
                self.store.adam_m.copy_(old.adam_m.to(device))
                self.store.adam_v.copy_(old.adam_v.to(device))
                self.store.step = old.step
        elif not self.store.attached():
            # someone replaced parameter storage (load_state_dict keeps views; .to()/.float() does not)
            with torch.no_grad():
                for p in self.store.params:
                    o = self.store.off(p)
                    self.store.flat_p[o:o + p.numel()].copy_(p.detach().reshape(-1).to(device=device, dtype=torch.float32))
            self.store.attach()
            self.store.touch()       # `p.data = ...` keeps the Parameters' version counters: the packed operands are stale anyway
        return self.store

    def plan_for(self, x, inference=False, need_input_grad=False):
        if x.ndim != self.net.dimensions + 2 or x.shape[1] != self.net.in_channels:
            nd = self.net.dimensions
            raise ValueError(f"expected input (B,{self.net.in_channels},{'H,W,D' if nd == 3 else 'H,W'}), got {tuple(x.shape)}")
        return self.plan_for_shape(x.device, x.shape[0], tuple(x.shape[2:]), inference, need_input_grad)

    def plan_for_shape(self, device, N, spatial, inference=False, need_input_grad=False):
        """``inference`` plans hold no backward buffers; an existing training plan of the same shape is reused instead"""
        self.ensure(device)
        nd = self.net.dimensions
        sp = tuple(spatial) + ((1,) if nd == 2 else ())
        nlev = len(self.net.channels) - 1
        for v in sp[:nd]:
            if v % (2 ** nlev):
                raise ValueError(f"spatial size {v} is not divisible by 2^{nlev} (the skip concat needs it, as in MONAI)")
        key = (N,) + sp
        tdt = self.dt if inference else self.train_dt()
        if tdt != self.dt:
            key = key + ("train-bf16",)     # an fp16 model's training plans (bf16 storage) are never reused for its fp16 inference
        if need_input_grad and not inference:
            plan = self.plans.get(key + ("dx",))
            if plan is None:
                plan = self.plans[key + ("dx",)] = self._record(N, sp, False, True, tdt)
            self.last_plan = plan
            return plan
        plan = self.plans.get(key) or self.plans.get(key + ("dx",)) or (self.plans.get(key + ("inference",)) if inference else None)
        if plan is None:
            plan = self.plans[key + (("inference",) if inference else ())] = self._record(N, sp, inference, False, tdt)
        self.last_plan = plan
        return plan

    def _record(self, N, sp, inference, need_input_grad=False, dt=None):
        """record a Plan; bf16 tensors of 9..12 channels (the class logits' neighbours) are laid out 12 wide when every pass
        that touches them can move such rows (ctseg_conv_narrow_ok / ctseg_wgrad_narrow_ok), 16 wide otherwise"""
        from . import engine as eng
        dt = self.dt if dt is None else dt
        if nat.is16(dt) and os.environ.get("CTSEG_NARROW_ROWS", "1") != "0":
            eng.NARROW_ROWS[0] = True
            try:
                return Plan(self, N, *sp, inference=inference, need_input_grad=need_input_grad, dt=dt)
            except eng.NarrowUnsupported:
                pass
            finally:
                eng.NARROW_ROWS[0] = False
        return Plan(self, N, *sp, inference=inference, need_input_grad=need_input_grad, dt=dt)

    # ---- raw (no autograd) API used by the native training step, bench and tests ----
    def forward(self, x):
        return self.plan_for(x).forward(x)

    def logits_view(self, plan=None):
        plan = plan or self.last_plan
        if not getattr(plan, "logits_current", True):
            # the last forward on this plan stopped before the logits convolution (fit_step(keep_logits=False): the cross-entropy
            # ran in that convolution's epilogue and the fp32 logits were never written): what the buffer holds is an older step's
            raise nat.NativeError("the logits of the last step were not materialised (fit_step(keep_logits=False) fuses the "
                                  "cross-entropy into the logits convolution); run the step with keep_logits=True, or a forward")
        v = plan.logits.valid()
        return v[..., 0] if self.net.dimensions == 2 else v

    def backward(self, plan=None, hooks=None):
        (plan or self.last_plan).backward(hooks)

    # ---- autograd surface (drop-in for loss.backward()) ----
    def forward_autograd(self, x):
        train = torch.is_grad_enabled() and any(p.requires_grad for p in self.net.parameters())
        plan = self.plan_for(x, inference=not train, need_input_grad=train and x.requires_grad)
        params = self.store.params
        if train:
            return _UNetFn.apply(x, self, plan, *params)
        plan.forward(x)
        return self.logits_view(plan).clone()


class _GradHandOff:
    """Gives autograd the parameter gradients WITHOUT a copy: the backward kernels write every gradient into the flat buffer, and
    ``p.grad`` becomes a (cached) view of it — no per-parameter clone, no AccumulateGrad add, and ``Adam.step`` (capstone_amd.optim)
    reads the flat buffer directly.  ``loss.backward()``'s contract is ``p.grad += dL/dp``, so what the parameters hold when the
    backward starts decides:
      * ``p.grad is None`` (``zero_grad()`` with torch's default ``set_to_none=True``, or the first step): ``p.grad`` = the view;
      * ``p.grad`` is that view already (``zero_grad(set_to_none=False)`` zeroed it in place, or a second backward without
        zero_grad = gradient accumulation): its content is saved before the kernels overwrite the buffer and added afterwards;
      * ``p.grad`` is some other tensor: the view is added to it."""

    def __init__(self, store):
        self.store = store
        views = getattr(store, "_grad_views", None)
        if views is None:
            views = store._grad_views = [store.grad_view(p) for p in store.params]
        self.views = views
        base = store.flat_g.data_ptr()
        self.kept = None
        self.aliased = []
        for i, p in enumerate(store.params):
            g = p.grad
            if g is not None and g.data_ptr() == base + 4 * store.off(p) and g.device == store.flat_g.device:
                self.aliased.append(i)
        if self.aliased:
            self.kept = store.flat_g.clone()

    def publish(self, scale=1.0):
        """after the backward program (and the data-parallel exchange) ran.  ``scale``: 1 / world of the gradient MEAN, applied to
        the flat buffer here so that ``p.grad`` is what DDP would have left there."""
        st = self.store
        if scale != 1.0:
            nat.call("ctseg_scale_inplace", st.flat_g.data_ptr(), F32, st.n, None, float(scale))
        al = set(self.aliased)
        for i, (p, v) in enumerate(zip(st.params, self.views)):
            if not p.requires_grad:
                continue
            if i in al:
                o = st.off(p)
                v.add_(self.kept[o:o + p.numel()].view(p.shape))
                p.grad = v
            elif p.grad is None:
                p.grad = v
            else:
                p.grad.add_(v)
        self.kept = None


def _backward_with_exchange(engine, plan):
    """run the recorded backward program; when the engine carries a gradient reducer (distributed.attach) the readiness hooks
    fire the chunked all-reduce from inside it.  Returns the scale that turns the summed flat gradient into the mean."""
    reducer = engine.reducer
    if reducer is not None:
        plan.backward(reducer.hooks(plan))
        return reducer.finish()
    engine.warn_if_unattached()
    plan.backward()
    return 1.0


class _StepLossFn(torch.autograd.Function):
    """The scalar loss of a training step whose forward ALREADY ran the loss and wrote d loss / d logits for an upstream gradient of
    1 into the plan (fused head: the logits convolution + cross-entropy in one launch; or the one-pass fused cross-entropy over
    materialised logits).  ``backward`` applies the actual upstream gradient (a launch that returns at once when it is 1), runs the
    recorded backward program (+ the data-parallel exchange when the module is attached) and hands the flat gradient to autograd
    without a copy.  This is what lets ``training_step -> loss.backward() -> optimizer.step()`` (the surface the reference's
    Lightning loop drives, capstone/volumetric/base_trainer.py:80-82,113-114) do the same GPU work as ``fit_step``."""

    @staticmethod
    def forward(ctx, loss_value, engine, plan, module, *params):
        ctx.engine, ctx.plan, ctx.gen, ctx.module = engine, plan, plan.fwd_gen, module
        return loss_value.detach().clone()

    @staticmethod
    def backward(ctx, g):
        engine, plan, module = ctx.engine, ctx.plan, ctx.module
        if plan.fwd_gen != ctx.gen:
            raise RuntimeError("the activations of this step were overwritten by a later forward on the same input shape "
                               "(one set of activation buffers per shape): call backward() before the next training forward")
        if not getattr(plan, "dlogits_is_current", False):
            raise RuntimeError("backward() ran twice on the same training step (its buffers are consumed by the first)")
        plan.dlogits_is_current = False
        dl = plan.dlogits.t
        gs = g.detach().to(device=dl.device, dtype=torch.float32).contiguous()
        nat.call("ctseg_scale_inplace", dl.data_ptr(), plan.dt, dl.numel(), gs.data_ptr(), 1.0)
        hand = _GradHandOff(engine.store)
        hand.publish(_backward_with_exchange(engine, plan))
        return (None, None, None, None) + (None,) * len(ctx.engine.store.params)


class _UNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, engine, plan, *params):
        """Returns a VIEW of the plan's logits buffer (the fused loss pass reads it in place, channels-last): a later forward
        on the same shape overwrites it, like the activations this node's backward needs — clone it to keep a prediction
        across steps.  ``backward`` refuses to run on overwritten activations (generation check) instead of returning
        gradients of the wrong batch."""
        plan.forward(x)
        ctx.engine, ctx.plan, ctx.gen = engine, plan, plan.fwd_gen
        return engine.logits_view(plan)

    @staticmethod
    def backward(ctx, g):
        engine, plan = ctx.engine, ctx.plan
        if plan.fwd_gen != ctx.gen:
            raise RuntimeError("the activations of this forward were overwritten by a later forward on the same input shape "
                               "(one set of activation buffers per shape): call backward() before the next training forward")
        fused = getattr(plan, "dlogits_is_current", False)
        if not fused:
            C = engine.net.out_channels
            gv = g if engine.net.dimensions == 3 else g.unsqueeze(-1)
            plan.dlogits.t[..., :C].copy_(gv.permute(0, 2, 3, 4, 1))
        plan.dlogits_is_current = False
        hand = _GradHandOff(engine.store)        # p.grad becomes a view of the flat gradient buffer: no per-parameter copies
        # data-parallel (distributed.attach): the same exchange as fit_step / _StepLossFn — every loss.backward() leaves the MEAN
        # gradient over the ranks in p.grad, as Lightning's DDP does (capstone/volumetric/base_trainer.py:196,217), whichever loss
        # recipe or entry point (training_step with Dice / Focal / GDL, forward() + a custom loss) led here
        hand.publish(_backward_with_exchange(engine, plan))
        grads = [None] * len(engine.store.params)
        gx = None
        if ctx.needs_input_grad[0]:
            if plan.dx is None:
                raise RuntimeError("this plan was recorded without an input-gradient pass")
            gx = plan.dx.valid().float()                       # (B, Cin, X, Y, Z) view of the channels-last gradient
            gx = (gx[..., 0] if engine.net.dimensions == 2 else gx).contiguous()
        return (gx, None, None, *grads)
