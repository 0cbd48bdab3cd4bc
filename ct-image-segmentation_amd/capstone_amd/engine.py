"""Host-side execution engine for the MI355X U-Net path.

Walks the MONAI-shaped module tree (capstone_amd.models.unet) ONCE per (device, dtype, input
shape) and records the forward and backward passes as flat programs of C-ABI calls
(include/ctseg_hip.h) over pre-allocated channels-last buffers.  Running a step is then a plain
loop over ``(cfunc, args)`` tuples on the current HIP stream: no autograd graph, no per-step
allocation, no tracing compiler.

What is fused where (see DESIGN.md for the byte accounting):
  * the stride-2 residual conv and ``unit0`` conv of a down ResidualUnit share their input and
    geometry -> ONE implicit GEMM with 2*C columns (and one fused dgrad / wgrad);
  * InstanceNorm statistics come out of the conv epilogue; ``prelu(norm(y)) + residual`` is one
    elementwise pass that writes straight into the skip-concat buffer (torch.cat is free);
  * ConvTranspose3d runs as 8 output-parity classes in one launch, no zero insertion; its input
    gradient is a stride-2 conv pass, the stride-2 conv's input gradient is the 8-class pass;
  * every gradient lands in one flat fp32 buffer (all-reduce + Adam are single launches).
"""
import ctypes as C
import math
import os

import numpy as np
import torch

from . import _native as nat
from ._native import BF16, F32


def rup(a, b):
    return (a + b - 1) // b * b


class Act:
    """Channels-last activation handle: tensor [N,X,Y,Z,ld], C valid channels starting at c0."""
    __slots__ = ("t", "C", "c0", "dt", "pending_norm", "bst")

    def __init__(self, t, C, c0=0, dt=None):
        self.t, self.C, self.c0 = t, C, c0
        self.pending_norm = None      # a _NormAct whose InstanceNorm + PReLU the consumers of this raw conv output apply on load
        # a GRADIENT tensor whose producing pass already took the backward statistics of the norm it feeds in its epilogue
        # (ctseg_conv_desc::bst_*): (norm, partials, P, ld) — that norm's backward then skips its reduce pass
        self.bst = None
        self.dt = dt if dt is not None else {torch.float32: F32, torch.float16: nat.F16}.get(t.dtype, BF16)

    @property
    def dims(self):
        return tuple(self.t.shape[:4])

    @property
    def ld(self):
        return self.t.shape[4]

    @property
    def S(self):
        s = self.t.shape
        return s[1] * s[2] * s[3]

    def ptr(self):
        return self.t.data_ptr() + self.c0 * self.t.element_size()

    def slice(self, c0, C):
        a = Act(self.t, C, self.c0 + c0, self.dt)
        b = self.bst
        if b is not None and b[0] == "slice" and b[1] == c0 and b[2] == C:
            a.bst = b[3]              # the column range whose backward statistics the producing pass took
        return a

    def valid(self):
        """torch view of the valid channels, logical NC[XYZ] order"""
        return self.t[..., self.c0:self.c0 + self.C].permute(0, 4, 1, 2, 3)


class SplitAct:
    """two dense tensors standing for one [.., c_split | ..] channel range (ctseg_conv_desc::out2): the consumers that used to
    take interleaved channel slices of one buffer — half of every cache line — get full-line tensors instead"""

    def __init__(self, a, b):
        self.a, self.b = a, b
        self.dims, self.C, self.dt = a.dims, a.C + b.C, a.dt

    def slice(self, c0, C):
        if c0 == 0 and C == self.a.C:
            return self.a
        if c0 == self.a.C and C == self.b.C:
            return self.b
        raise ValueError("a split activation can only be taken apart at its split point")


class NarrowUnsupported(Exception):
    """a pass of the plan being built cannot move 12-wide rows: the plan is rebuilt with 16-byte chunked rows"""


NARROW_ROWS = [False]     # set while a Plan is being recorded (Engine.plan_for_shape)


def default_ld(C, dt):
    """channel stride of a fresh activation: 16-byte chunks; bf16 tensors of 9..12 channels 12 wide while NARROW_ROWS"""
    if NARROW_ROWS[0] and nat.is16(dt) and rup(C, 4) == 12:
        return 12
    return rup(C, nat.epc(dt))


def new_act(N, X, Y, Z, C, dt, device, ld=None, zero=True):
    ld = ld if ld is not None else default_ld(C, dt)
    f = torch.zeros if zero else torch.empty
    return Act(f((N, X, Y, Z, ld), dtype=nat.torch_dtype(dt), device=device), C, 0, dt)


# ------------------------------------------------------------------------------------------------
# taps
# ------------------------------------------------------------------------------------------------
def _pack_off(dx, dy, dz):
    return (dx & 255) | ((dy & 255) << 8) | ((dz & 255) << 16)


def _tap_product(axes, k, dims):
    """axes: per spatial axis a list of (kernel index t, offset d). Returns [(torch tap id, packed offset)]."""
    out = []
    if dims == 3:
        for tx, dx in axes[0]:
            for ty, dy in axes[1]:
                for tz, dz in axes[2]:
                    out.append(((tx * k + ty) * k + tz, _pack_off(dx, dy, dz)))
    else:
        for tx, dx in axes[0]:
            for ty, dy in axes[1]:
                out.append((tx * k + ty, _pack_off(dx, dy, 0)))
    return out


def classes_plain(k, dims, off):
    """one class; off(t) gives the gather offset of kernel index t"""
    ax = [[(t, off(t)) for t in range(k)]] * dims
    return [((0, 0, 0), _tap_product(ax, k, dims))]


def classes_up(k, dims):
    """stride-2 'transposed' pass (ConvTranspose fwd, stride-2 conv dgrad): one class per output parity.
    written x = 2r + p = 2*xi - 1 + t  =>  p=0: t=1, xi=r ; p=1: t=0, xi=r+1 | t=2, xi=r"""
    assert k == 3
    per = {0: [(1, 0)], 1: [(0, 1), (2, 0)]}
    out = []
    rz = (0, 1) if dims == 3 else (0,)
    for px in (0, 1):
        for py in (0, 1):
            for pz in rz:
                ax = [per[px], per[py]] + ([per[pz]] if dims == 3 else [])
                out.append(((px, py, pz), _tap_product(ax, k, dims)))
    return out


# ------------------------------------------------------------------------------------------------
# parameters: one flat fp32 buffer (+ flat grad) ordered by gradient readiness in backward
# ------------------------------------------------------------------------------------------------
class ParamStore:
    def __init__(self, params, device):
        self.params = list(params)
        self.device = device
        self.offsets, off = {}, 0
        for p in self.params:
            self.offsets[id(p)] = off
            off += p.numel()
        self.n = off
        self.n_pad = rup(off, 4)
        self.zero_index = self.n_pad  # flat_p[n_pad:] stays 0.0 forever (pad source for packing)
        self.flat_p = torch.zeros(self.n_pad + 4, dtype=torch.float32, device=device)
        self.flat_g = torch.zeros(self.n_pad, dtype=torch.float32, device=device)
        self.adam_m = self.adam_v = None
        self.step = 0
        # bumped by every write to flat_p that bypasses the Parameters' own version counters (ctseg_adam_step writes through raw
        # pointers; broadcasts and re-attach copies write the flat buffer): EVERY plan's Packer compares it, so a plan of another
        # shape (validation, sliding window, a short last batch) never runs on packed weights from before the update
        self.generation = 0
        with torch.no_grad():
            for p in self.params:
                o = self.offsets[id(p)]
                self.flat_p[o:o + p.numel()].copy_(p.detach().reshape(-1).to(device=device, dtype=torch.float32))
        self.attach()

    def attach(self):
        """make every Parameter a view of the flat buffer"""
        for p in self.params:
            o = self.offsets[id(p)]
            p.data = self.flat_p[o:o + p.numel()].view(p.shape)

    def attached(self):
        base = self.flat_p.data_ptr()
        return all(p.data_ptr() == base + 4 * self.offsets[id(p)] and p.device == self.flat_p.device for p in self.params)

    def off(self, p):
        return self.offsets[id(p)]

    def p_ptr(self, p):
        return self.flat_p.data_ptr() + 4 * self.offsets[id(p)]

    def g_ptr(self, p):
        return self.flat_g.data_ptr() + 4 * self.offsets[id(p)]

    def grad_view(self, p):
        o = self.offsets[id(p)]
        return self.flat_g[o:o + p.numel()].view(p.shape)

    def version(self):
        return (self.generation, sum(p._version for p in self.params))

    def touch(self):
        """flat_p was rewritten behind the Parameters' backs: every plan's packed operands are stale"""
        self.generation += 1

    def ensure_adam_state(self):
        if self.adam_m is None:
            self.adam_m = torch.zeros_like(self.flat_g)
            self.adam_v = torch.zeros_like(self.flat_g)

    def adam_step(self, lr, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0):
        """torch.optim.Adam semantics (capstone/volumetric/base_trainer.py:113-114), one launch."""
        self.ensure_adam_state()
        self.step += 1
        nat.call("ctseg_adam_step", self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.adam_m.data_ptr(),
                 self.adam_v.data_ptr(), self.n, lr, betas[0], betas[1], eps, self.step, grad_scale)
        self.touch()


# ------------------------------------------------------------------------------------------------
# one implicit-GEMM layer = one conv module, or a fused pair sharing input and geometry
# ------------------------------------------------------------------------------------------------
class GemmLayer:
    def __init__(self, plan, name, transposed, k, stride, cin, parts, cg, need_dgrad=True):
        """parts: [(weight Parameter, bias Parameter, cout)], cg: gathered channel stride of the input"""
        self.plan, self.name, self.transposed, self.k, self.s = plan, name, transposed, k, stride
        self.cin, self.parts, self.cg = cin, parts, cg
        self.Cn = sum(c for _, _, c in parts)
        self.dims = plan.dims
        self.T = k ** self.dims
        dt = plan.dt
        e = nat.epc(dt)
        self.cgd = rup(self.Cn, e)          # gathered stride when dY is the gathered tensor (dgrad / convT wgrad)
        p = (k - 1) // 2
        if not transposed:
            self.fwd_classes = classes_plain(k, self.dims, lambda t: t - p)
            self.dg_classes = classes_plain(k, self.dims, lambda t: p - t) if stride == 1 else classes_up(k, self.dims)
        else:
            assert len(parts) == 1
            self.fwd_classes = classes_up(k, self.dims) if stride == 2 else classes_plain(k, self.dims, lambda t: p - t)
            self.dg_classes = classes_plain(k, self.dims, lambda t: t - p)
        self.wg_taps = classes_plain(k, self.dims, lambda t: t - p)[0][1]
        assert [t for t, _ in self.wg_taps] == list(range(self.T))
        self.fwd_pack = plan.packer.add(self, "fwd")
        self.dg_pack = plan.packer.add(self, "dgrad") if need_dgrad else None  # the stem needs no input gradient
        self.bias_off = plan.packer.add_bias(self)
        self.x_dims = None

    # ---- index math for the packer: flat-parameter index of W element for (GEMM row n, gathered channel g, tap id t)
    def src_index(self, mode, n, g, t):
        st = self.plan.store
        cin, T = self.cin, self.T
        idx = np.full(np.broadcast(n, g, t).shape, st.zero_index, dtype=np.int64)
        if not self.transposed:
            r0 = 0
            for w, _, cout in self.parts:
                o = st.off(w)
                if mode == "fwd":    # rows = out channels, gather = in channels
                    m = (n >= r0) & (n < r0 + cout) & (g < cin)
                    v = o + ((n - r0) * cin + g) * T + t
                else:                # dgrad: rows = in channels, gather = dY channels
                    m = (g >= r0) & (g < r0 + cout) & (n < cin)
                    v = o + ((g - r0) * cin + n) * T + t
                idx = np.where(m, v, idx)
                r0 += cout
        else:
            w, _, cout = self.parts[0]
            o = st.off(w)
            if mode == "fwd":        # weight [cin][cout][T]: rows = cout, gather = cin
                m = (n < cout) & (g < cin)
                v = o + (g * cout + n) * T + t
            else:
                m = (n < cin) & (g < cout)
                v = o + (n * cout + g) * T + t
            idx = np.where(m, v, idx)
        return idx

    def src_parts(self, mode):
        """the same mapping as src_index, structured (ctseg_pack_part): [(o, n_lo, n_hi, g_lo, g_hi, SN, SG)] with
        source(n, g, t) = o + (n - n_lo) * SN + (g - g_lo) * SG + t for (n, g) inside the ranges"""
        st, cin, T = self.plan.store, self.cin, self.T
        out = []
        if not self.transposed:
            r0 = 0
            for w, _, cout in self.parts:
                o = st.off(w)
                out.append((o, r0, r0 + cout, 0, cin, cin * T, T) if mode == "fwd" else (o, 0, cin, r0, r0 + cout, T, cin * T))
                r0 += cout
        else:
            w, _, cout = self.parts[0]
            o = st.off(w)
            out.append((o, 0, cout, 0, cin, T, cout * T) if mode == "fwd" else (o, 0, cin, 0, cout, cout * T, T))
        return out

    def rows_gather(self, mode):
        """(real GEMM rows, gathered channel stride) for a pack mode"""
        return (self.Cn, self.cg) if mode == "fwd" else (self.cin, self.cgd)

    # ---- geometry ---------------------------------------------------------------------------------
    def out_dims(self, xd):
        N, X, Y, Z = xd
        p = (self.k - 1) // 2
        if self.transposed:
            f = (lambda v: v * 2) if self.s == 2 else (lambda v: v)
        else:
            f = lambda v: (v + 2 * p - self.k) // self.s + 1
        return (N, f(X), f(Y), f(Z) if self.dims == 3 else 1)

    def _desc(self, pack, classes, gathered, out, rowgrid, sin, sout, Cn, cg, bias_ptr, add, stats, out_f32, out2=None):
        plan = self.plan
        d = nat.ConvDesc()
        d.in_, d.w, d.bias, d.out = gathered.ptr(), plan.packer.ptr(pack), bias_ptr, out.ptr()
        if out2 is not None:
            d.out2, d.out2_col0, d.o2_ld = out2.ptr(), out.C, out2.ld
        d.add = add.ptr() if add is not None else None
        d.dtype = plan.dt
        d.N, d.Xi, d.Yi, d.Zi = gathered.dims
        d.Xr, d.Yr, d.Zr = rowgrid
        _, d.Xo, d.Yo, d.Zo = out.dims
        d.Cg, d.Cn = cg, Cn
        narrow_out = nat.is16(plan.dt) and not out_f32 and out.ld == 12
        d.Cn_store = rup(Cn, 4 if (out_f32 or narrow_out) else nat.epc(plan.dt))
        assert out2 is not None or out.c0 + d.Cn_store <= out.ld, (self.name, out.c0, d.Cn_store, out.ld)
        d.g_ld, d.o_ld = gathered.ld, out.ld
        d.add_ld = add.ld if add is not None else 0
        self._narrow = nat.is16(plan.dt) and (narrow_out or gathered.ld == 12 or
                                           (add is not None and add.t.dtype != torch.float32 and add.ld == 12))
        d.sin, d.sout = sin, sout
        d.out_f32 = 1 if out_f32 else 0
        d.add_f32 = 1 if (add is not None and add.t.dtype == torch.float32 and plan.dt != F32) else 0
        d.nclass = len(classes)
        bk = 128 // nat.elsize(plan.dt)
        for i, ((ox, oy, oz), taps) in enumerate(classes):
            c = d.cls[i]
            c.ntaps, c.kpad, c.w_off = len(taps), pack["kpads"][i], pack["w_offs"][i]
            c.ox, c.oy, c.oz = ox, oy, oz
            for j, (_, off) in enumerate(taps):
                c.taps[j] = off
            assert c.kpad % bk == 0
        if stats is not None:
            d.stats, d.stats_ld, d.stats_tiles, d.stats_tile0 = stats.partials.data_ptr(), stats.ld, stats.tiles, 0
        if self._narrow and nat.query("ctseg_conv_narrow_ok", d) != 1:
            raise NarrowUnsupported(self.name)
        return d

    def _try_split(self, d, od, split_at):
        """two dense outputs instead of one [split_at | rest] buffer where the kernel taking this pass can (stem, stride-2 halo)"""
        if split_at is None or os.environ.get("CTSEG_SPLIT_OUT", "1") == "0":
            return None
        d.out2_col0 = split_at
        if nat.query("ctseg_conv_split_ok", d) != 1:
            d.out2_col0 = 0
            return None
        plan = self.plan
        a = new_act(*od, split_at, plan.dt, plan.device)
        b = new_act(*od, d.Cn - split_at, plan.dt, plan.device)
        d.out, d.o_ld = a.ptr(), a.ld
        d.out2, d.o2_ld = b.ptr(), b.ld
        return SplitAct(a, b)

    def emit_fwd(self, x, out=None, want_stats=False, add=None, out_f32=False, split_at=None):
        plan = self.plan
        self.x_dims = x.dims
        od = self.out_dims(x.dims)
        own_out = out is None
        if out is None:
            dt = F32 if out_f32 else plan.dt
            out = new_act(*od, self.Cn, dt, plan.device)
        assert out.dims == od and out.C == self.Cn, (self.name, out.dims, od)
        if self.transposed and self.s == 2:
            rowgrid, sin, sout = x.dims[1:], 1, 2
        else:
            rowgrid, sin, sout = od[1:], (1 if self.transposed else self.s), 1
        stats = None
        bias_ptr = plan.packer.bias_ptr(self.bias_off)
        pend = getattr(x, "pending_norm", None)
        d = self._desc(self.fwd_pack, self.fwd_classes, x, out, rowgrid, sin, sout, self.Cn, self.cg, bias_ptr, add, None, out_f32)
        if pend is not None:
            # the producing layer's InstanceNorm + PReLU is applied to the operand on its way into this pass (the activation is
            # never written) where the kernel taking the pass can; otherwise the apply pass is recorded now
            assert add is None or add is x, "a deferred norm feeds the pass and (optionally) its identity residual only"
            d.in_mean_rstd, d.in_alpha, d.in_norm_C = pend.mr.data_ptr(), plan.store.p_ptr(pend.alpha), x.C
            if nat.query("ctseg_conv_in_norm_ok", d) != 1:
                xm = pend.materialise(x)
                d = self._desc(self.fwd_pack, self.fwd_classes, xm, out, rowgrid, sin, sout, self.Cn, self.cg, bias_ptr,
                               xm if add is not None else None, None, out_f32)
                x, add = xm, (xm if add is not None else None)
        if want_stats:
            tiles = nat.lib().ctseg_conv_num_tiles(d)
            assert tiles > 0
            stats = NormStats(plan, od[0], tiles, self.Cn, od[1] * od[2] * od[3])
            d.stats, d.stats_ld, d.stats_tiles, d.stats_tile0 = stats.partials.data_ptr(), stats.ld, stats.tiles, 0
        if own_out and add is None and not out_f32:
            sp = self._try_split(d, od, split_at)
            if sp is not None:
                out = sp
        plan.emit("ctseg_conv_igemm", d, keep=(x, out, add, stats))
        return out, stats

    def _try_bst(self, d, out, norm, col0=0):
        """ask the pass that writes the gradient ``out`` to take the backward statistics of ``norm`` (the InstanceNorm + PReLU whose
        output gradient ``out`` is, channels [col0, col0 + C) of the pass) in its epilogue; returns the tensor the statistics
        belong to (``out``, or the second half of a split output) or None when the kernel taking the pass cannot"""
        if norm is None or os.environ.get("CTSEG_BST", "1") == "0":
            return None       # (the library declines fp32 storage: its trajectory tests pin a summation order)
        y = norm.y
        tgt = out.slice(col0, y.C) if (col0 or isinstance(out, SplitAct)) else out
        if tgt.C != y.C or tgt.dims != y.dims:
            return None
        d.bst_y, d.bst_y_ld, d.bst_C, d.bst_col0 = y.ptr(), y.ld, y.C, col0
        P = nat.query("ctseg_conv_bwd_stats_slots", d)
        if P <= 0:
            d.bst_y, d.bst_y_ld, d.bst_C, d.bst_col0 = None, 0, 0, 0
            return None
        plan = self.plan
        ld = rup(y.C, 4)
        part = torch.zeros((y.dims[0], P, 3, ld), dtype=torch.float32, device=plan.device)
        d.bst_mean_rstd, d.bst_alpha = norm.mr.data_ptr(), plan.store.p_ptr(norm.alpha)
        d.bst_partials, d.bst_P, d.bst_ld = part.data_ptr(), P, ld
        tgt.bst = (norm, part, P, ld)
        if tgt is not out and not isinstance(out, SplitAct):
            out.bst = ("slice", col0, y.C, tgt.bst)      # slices of a plain Act are made afresh: the parent remembers the mark
        return tgt

    def emit_dgrad(self, dy, out=None, add=None, split_at=None, bst=None, bst_col0=0):
        """input gradient: gathered = dY (Cn channels), written = dX (cin channels).  ``bst``: the _NormAct whose backward consumes
        the written gradient (its channels [bst_col0, ..)): its statistics are taken in this pass's epilogue where the kernel can"""
        plan = self.plan
        assert self.dg_pack is not None, f"{self.name}: built without an input-gradient operand"
        xd = self.x_dims
        own_out = out is None
        if out is None:
            out = new_act(*xd, self.cin, plan.dt, plan.device)
        assert out.dims == xd and out.C == self.cin and dy.C == self.Cn
        if self.transposed and self.s == 2:       # strided conv over dOut
            rowgrid, sin, sout = xd[1:], 2, 1
        elif (not self.transposed) and self.s == 2:  # 8-class pass over dY
            assert all(a == 2 * b for a, b in zip(xd[1:1 + self.dims], dy.dims[1:1 + self.dims])), \
                "stride-2 convolutions need even input sizes (as MONAI's UNet does for the skip concat)"
            rowgrid, sin, sout = dy.dims[1:], 1, 2
        else:
            rowgrid, sin, sout = xd[1:], 1, 1
        d = self._desc(self.dg_pack, self.dg_classes, dy, out, rowgrid, sin, sout, self.cin, self.cgd, None, add, None, False)
        if own_out and add is None:
            sp = self._try_split(d, xd, split_at)
            if sp is not None:
                out = sp
        marked = self._try_bst(d, out, bst, bst_col0)
        plan.emit("ctseg_conv_igemm", d, keep=(dy, out, add, marked.bst[1] if marked is not None else None, bst.y if marked is not None else None))
        return out

    def wgrad_dyn_ok(self, x, dy_lo, y):
        """can this layer's weight-gradient pass form the upper columns of dY on load (ctseg_wgrad_desc::dyn_*)?  ``dy_lo``: the
        gradient of the lower column block, ``y``: the forward output of the upper one (the norm's input)"""
        if self.transposed or self.Cn != dy_lo.C + y.C or os.environ.get("CTSEG_WGRAD_DYN", "1") == "0":
            return False
        d = nat.WgradDesc()
        d.in_, d.dy, d.dtype = x.ptr(), dy_lo.ptr(), self.plan.dt
        d.N, d.Xi, d.Yi, d.Zi = x.dims
        d.Xr, d.Yr, d.Zr = y.dims[1:]
        d.Cg, d.Cn, d.g_ld, d.d_ld, d.sin, d.ntaps = self.cg, self.Cn, x.ld, dy_lo.ld, self.s, self.T
        for j, (_, off) in enumerate(self.wg_taps):
            d.taps[j] = off
        d.splits, d.kpad_w, d.cn_pad = 1, rup(self.T * self.cg + 1, 128), rup(self.Cn, nat.lib().ctseg_wgrad_tile_cols(self.Cn))
        d.dyn_col0, d.dyn_g, d.dyn_y = dy_lo.C, y.ptr(), y.ptr()
        d.dyn_g_ld, d.dyn_y_ld = rup(y.C, nat.epc(self.plan.dt)), y.ld
        return nat.query("ctseg_wgrad_dy_norm_ok", d) == 1

    def _wgrad_splits(self, d, N, rows, on_load):
        """row ranges per sample of a split-K weight-gradient pass.  The library says how many workgroups one slab takes and how
        many a CU holds (ctseg_conv_wgrad_wgs_per_slab); ``on_load``: the descriptor will carry dyn_* / in_norm fields (generic
        kernel whatever the query says for the plain descriptor)"""
        plan, lib = self.plan, nat.lib()
        per_cu, stage_bytes = C.c_int32(0), C.c_int32(0)
        wps = lib.ctseg_conv_wgrad_wgs_per_slab(C.byref(d), C.byref(per_cu), C.byref(stage_bytes))
        m = 8 // math.gcd(N, 8)       # slabs (N * splits) in multiples of 8: all tiles of a slab then run on one XCD (one L2 fetch of its rows)
        if wps > 0 and per_cu.value == 1 and not on_load and "CTSEG_WGRAD_TARGET_WGS" not in os.environ:
            # one 512-thread workgroup per CU.  Cost of a split count s (microseconds, constants measured on the MI355X):
            #   rounds of 256 workgroups x stages per workgroup x time of a stage (its staging bytes at ~60 GB/s per CU, never below 0.25 us)
            #   + a fixed ~6 us per round (ring fill, accumulator store)
            #   + the fp32 slabs, written here and read back by the reduce, at ~3 TB/s each way
            slab_bytes = d.kpad_w * d.cn_pad * 4
            t_stage = max(stage_bytes.value / 60e3, 0.25)
            best = None
            for s_ in range(1, max(1, min(rows // 192, 256)) + 1):
                wgs = wps * N * s_
                stages = max(8, math.ceil(math.ceil(rows / s_) / 32))
                cost = math.ceil(wgs / 256.0) * (stages * t_stage + 6.0) + 2.0 * N * s_ * slab_bytes / 3e6
                if (N * s_) % 8 != 0 and wps > 1:
                    cost *= 1.07      # tiles of a slab spread over the XCDs: every L2 fetches its own copy of the rows
                if best is None or cost < best[0]:
                    best = (cost, s_)
            return best[1]
        nwg = ((d.kpad_w // 128) * (d.cn_pad // lib.ctseg_wgrad_tile_cols(d.Cn))) * N
        # workgroups to aim for: 1024 with 16-bit storage (2048 measured +0.1 ms/step once the kernel's address code got cheaper:
        # more slabs to write and reduce; 512 is +0.35); fp32 storage keeps 2048 (its trajectory test pins a summation order)
        target = int(os.environ.get("CTSEG_WGRAD_TARGET_WGS", "1024" if nat.is16(plan.dt) else "2048"))
        splits = max(1, min(math.ceil(target / nwg), math.ceil(rows / 512), 1024))
        if splits >= m:
            splits = splits // m * m
        return splits

    def emit_wgrad(self, x, dy, bias_done=False, dyn=None):
        """weight + bias gradients straight into the flat gradient buffer (deterministic split-K).  ``dyn`` = (g, norm, sums): ``dy``
        holds the lower column block only; the upper one is the InstanceNorm + PReLU backward of ``g`` through ``norm``, formed on
        load by the pass (wgrad_dyn_ok)."""
        plan, st = self.plan, self.plan.store
        lib = nat.lib()
        if dyn is not None:
            assert not self.transposed
            gathered, dyy, cg, A, cn = x, dy, self.cg, self.cin, self.Cn
            rowgrid, sin = dyn[1].y.dims[1:], self.s
            assert dy.C + dyn[0].C == cn and dyn[1].y.C == dyn[0].C
        elif not self.transposed:
            gathered, dyy, cg, A, cn = x, dy, self.cg, self.cin, self.Cn
            rowgrid, sin = dy.dims[1:], self.s
        else:
            assert self.s == 2, "stride-1 transposed conv wgrad not implemented"
            gathered, dyy, cg, A, cn = dy, x, self.cgd, self.Cn, self.cin
            rowgrid, sin = x.dims[1:], 2
        assert dyn is not None or dyy.C == cn, (self.name, dyy.C, cn)
        N = gathered.dims[0]
        rows = rowgrid[0] * rowgrid[1] * rowgrid[2]
        bnw = lib.ctseg_wgrad_tile_cols(cn)
        kpad_w, cn_pad = rup(self.T * cg + 1, 128), rup(cn, bnw)
        d = nat.WgradDesc()
        d.in_, d.dy, d.dtype = gathered.ptr(), dyy.ptr(), plan.dt
        d.N, d.Xi, d.Yi, d.Zi = gathered.dims
        d.Xr, d.Yr, d.Zr = rowgrid
        d.Cg, d.Cn, d.g_ld, d.d_ld, d.sin = cg, cn, gathered.ld, dyy.ld, sin
        d.ntaps = self.T
        for j, (_, off) in enumerate(self.wg_taps):
            d.taps[j] = off
        d.splits, d.kpad_w, d.cn_pad = 1, kpad_w, cn_pad
        splits = self._wgrad_splits(d, N, rows, dyn is not None or getattr(gathered, "pending_norm", None) is not None)
        d.splits = splits
        if dyn is not None:
            g_up, norm, sums = dyn
            d.dyn_col0, d.dyn_g, d.dyn_g_ld, d.dyn_y, d.dyn_y_ld = dy.C, g_up.ptr(), g_up.ld, norm.y.ptr(), norm.y.ld
            d.dyn_mean_rstd, d.dyn_alpha, d.dyn_sums = norm.mr.data_ptr(), st.p_ptr(norm.alpha), sums.data_ptr()
            assert nat.query("ctseg_wgrad_dy_norm_ok", d) == 1, self.name
        pend = getattr(gathered, "pending_norm", None)
        if pend is not None:
            assert not self.transposed
            d.in_mean_rstd, d.in_alpha, d.in_norm_C = pend.mr.data_ptr(), st.p_ptr(pend.alpha), gathered.C
            if nat.query("ctseg_wgrad_in_norm_ok", d) != 1:
                gathered = pend.materialise(gathered)
                d.in_mean_rstd = d.in_alpha = None
                d.in_norm_C = 0
                d.in_, d.g_ld = gathered.ptr(), gathered.ld
        if plan.dt == BF16 and (gathered.ld == 12 or dyy.ld == 12) and nat.query("ctseg_wgrad_narrow_ok", d) != 1:
            raise NarrowUnsupported(self.name + " (weight gradient)")
        nslabs = lib.ctseg_conv_wgrad_slabs(d)     # N*splits, or one per persistent workgroup (LDS-halo kernel)
        assert nslabs > 0
        ws = torch.zeros(nslabs * kpad_w * cn_pad, dtype=torch.float32, device=plan.device)
        d.ws = ws.data_ptr()
        plan.emit("ctseg_conv_wgrad", d, keep=(gathered, dyy, ws, dyn))
        if not self.transposed:
            col0 = 0
            for w, b, cout in self.parts:
                plan.emit("ctseg_conv_wgrad_reduce", ws.data_ptr(), nslabs, kpad_w, cn_pad, A, cg, self.T, col0, cout,
                          st.g_ptr(w), st.g_ptr(b) if b is not None else None)
                col0 += cout
        else:
            w, b, cout = self.parts[0]
            # R[(t, co)][ci] -> W_T[ci][co][t]; the bias gradient of a transposed conv is sum over dOut: separate pass
            plan.emit("ctseg_conv_wgrad_reduce", ws.data_ptr(), nslabs, kpad_w, cn_pad, A, cg, self.T, 0, cn,
                      st.g_ptr(w), None)
            if b is not None and not bias_done:
                plan.emit_colsum(dy, st.g_ptr(b))


class NormStats:
    def __init__(self, plan, N, tiles, C, count):
        self.plan, self.N, self.tiles, self.C, self.count = plan, N, tiles, C, count
        self.ld = rup(C, 256 if C > 128 else nat.lib().ctseg_conv_tile_cols(C))   # 192x256 tile for C > 128
        self.partials = torch.zeros((N, tiles, 2, self.ld), dtype=torch.float32, device=plan.device)
        self.scratch = torch.zeros(N * 64 * 2 * self.ld + N, dtype=torch.float64, device=plan.device)   # + N completion counters

    def emit_finalize(self, col0, C, eps=1e-5):
        mr = torch.zeros((self.N, C, 2), dtype=torch.float32, device=self.plan.device)
        self.plan.emit("ctseg_instnorm_finalize", self.partials.data_ptr(), self.N, self.tiles, self.ld, col0, C,
                       float(self.count), float(eps), self.scratch.data_ptr(), mr.data_ptr(), keep=(self, mr))
        return mr


# ------------------------------------------------------------------------------------------------
# packed operands: one gather per optimizer step rebuilds every K-contiguous weight block
# ------------------------------------------------------------------------------------------------
class Packer:
    def __init__(self, plan):
        self.plan = plan
        self.blocks = []      # (offset, np index array)
        self.descs = []       # the same blocks as ctseg_pack_block descriptions
        self.total = 0
        self.bias_blocks, self.bias_total = [], 0
        self.buf = self.idx = self.bias_buf = self.bias_idx = None
        self.dirty = True
        self.version = None

    def add(self, layer, mode):
        dt = self.plan.dt
        bk = 128 // nat.elsize(dt)
        zero = self.plan.store.zero_index
        classes = layer.fwd_classes if mode == "fwd" else layer.dg_classes
        rows, gs = layer.rows_gather(mode)
        rows_pad = rup(rows, 256 if rows > 128 else 128)   # a column tile never reads past the padded rows
        info = {"kpads": [], "w_offs": [], "base": self.total}
        for _, taps in classes:
            nt = len(taps)
            kpad = rup(nt * gs, bk)
            n_, t_, g_ = np.broadcast_arrays(np.arange(rows_pad)[:, None, None],
                                             np.array([t for t, _ in taps])[None, :, None],
                                             np.arange(gs)[None, None, :])
            src = np.where(n_ < rows, layer.src_index(mode, n_, g_, t_), zero)
            blk = np.full((rows_pad, kpad), zero, dtype=np.int64)
            blk[:, :nt * gs] = src.reshape(rows_pad, nt * gs)
            info["kpads"].append(kpad)
            info["w_offs"].append(self.total - info["base"])
            self.blocks.append((self.total, blk.reshape(-1)))
            # the same block, structured (ctseg_pack_weights): checked against the index above element by element
            parts = layer.src_parts(mode)
            desc = {"dst_off": self.total, "kpad": kpad, "gs": gs, "ntaps": nt, "T": layer.T, "taps": [t for t, _ in taps],
                    "parts": parts, "rows": rows}
            if os.environ.get("CTSEG_PACK_CHECK", "1") != "0":
                chk = np.full((rows_pad, kpad), zero, dtype=np.int64)
                for (o, n_lo, n_hi, g_lo, g_hi, SN, SG) in parts:
                    nn, gg = np.arange(n_lo, min(n_hi, rows)), np.arange(g_lo, min(g_hi, gs))
                    for j, t in enumerate(desc["taps"]):
                        chk[n_lo:n_lo + len(nn), j * gs + g_lo:j * gs + g_lo + len(gg)] = \
                            o + (nn[:, None] - n_lo) * SN + (gg[None, :] - g_lo) * SG + t
                assert np.array_equal(chk, blk), f"{layer.name} {mode}: structured pack description disagrees with the index"
            self.descs.append(desc)
            self.total += rows_pad * kpad
        info["size"] = self.total - info["base"]
        return info

    def add_bias(self, layer):
        st = self.plan.store
        off = self.bias_total
        idx = []
        for _, b, cout in layer.parts:
            idx.append(np.arange(cout) + st.off(b) if b is not None else np.full(cout, st.zero_index))
        idx = np.concatenate(idx)
        pad = rup(len(idx), 4)
        full = np.full(pad, st.zero_index, dtype=np.int64)
        full[:len(idx)] = idx
        self.bias_blocks.append((off, full))
        self.bias_total += pad
        return off

    def ptr(self, info):
        assert self.buf is not None, "packer.finalize() must run before programs are recorded"
        return self.buf.data_ptr() + info["base"] * nat.elsize(self.plan.dt)

    def bias_ptr(self, off):
        assert self.bias_buf is not None
        return self.bias_buf.data_ptr() + 4 * off

    def finalize(self):
        dev = self.plan.device
        self.buf = torch.zeros(max(self.total, 8), dtype=nat.torch_dtype(self.plan.dt), device=dev)
        self.bias_buf = torch.zeros(max(self.bias_total, 4), dtype=torch.float32, device=dev)
        idx = np.full(self.total, self.plan.store.zero_index, dtype=np.int32)
        for off, blk in self.blocks:
            idx[off:off + len(blk)] = blk
        self.idx = torch.from_numpy(idx).to(dev)
        bidx = np.full(max(self.bias_total, 4), self.plan.store.zero_index, dtype=np.int32)
        for off, blk in self.bias_blocks:
            bidx[off:off + len(blk)] = blk
        self.bias_idx = torch.from_numpy(bidx).to(dev)
        self.n_idx, self.n_bidx = self.total, max(self.bias_total, 4)
        self.dirty = True
        # structured re-layout (ctseg_pack_weights) where every block's source region fits its LDS staging buffer
        self.pack_blocks = self.pack_rows = None
        fits = all(sum((min(gh, d["gs"]) - gl) * d["T"] for (_, _, _, gl, gh, _, _) in d["parts"]) <= nat.PACK_LDS_FLOATS and
                   len(d["parts"]) <= 2 for d in self.descs)
        if self.descs and fits and os.environ.get("CTSEG_PACK_STRUCTURED", "1") != "0":
            arr = (nat.PackBlock * len(self.descs))()
            rows = []
            for b, d in enumerate(self.descs):
                B = arr[b]
                B.dst_off, B.kpad, B.gs, B.ntaps, B.T, B.nparts = d["dst_off"], d["kpad"], d["gs"], d["ntaps"], d["T"], len(d["parts"])
                for j, t in enumerate(d["taps"]):
                    B.tap[j] = t
                for k, (o, n_lo, n_hi, g_lo, g_hi, SN, SG) in enumerate(d["parts"]):
                    P = B.part[k]
                    P.o, P.n_lo, P.n_hi, P.g_lo, P.g_hi, P.SN, P.SG = o, n_lo, n_hi, g_lo, min(g_hi, d["gs"]), SN, SG
                rows += [(b, n) for n in range(d["rows"])]
            import ctypes
            raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone()
            self.pack_blocks = raw.to(dev)
            # rows of the blocks of the FIRST forward pass first (Packer.first = its pack): refresh(part="first") rebuilds them alone,
            # the next step's first convolution waits for nothing else
            lo, hi = self.first if getattr(self, "first", None) else (0, 0)
            head = [(b, n) for (b, n) in rows if lo <= self.descs[b]["dst_off"] < hi]
            rest = [(b, n) for (b, n) in rows if not (lo <= self.descs[b]["dst_off"] < hi)]
            self.n_first_rows = len(head)
            self.pack_rows = torch.tensor(head + rest, dtype=torch.int32).reshape(-1).to(dev)
            self.n_pack_blocks, self.n_pack_rows = len(self.descs), len(rows)

    def can_split(self):
        """True when refresh(part="first") / refresh(part="rest") can rebuild the first forward pass's operands on their own"""
        return self.pack_blocks is not None and getattr(self, "n_first_rows", 0) > 0

    def stale(self):
        """True when refresh() (no force) would rebuild the operands: the parameters changed since the last complete re-layout"""
        return self.dirty or self.plan.store.version() != self.version

    def refresh(self, force=False, part=None):
        """part: None = everything; "first" = the biases and the packed operands of the first forward pass; "rest" = the others
        (the pair, in that order, equals None)"""
        st = self.plan.store
        ver = st.version()
        if not (force or self.dirty or ver != self.version):
            return
        if self.pack_blocks is not None:
            # (padding rows / K slots beyond ntaps * gs of a real row: the former keep the zeros of the allocation, the latter are
            # written as zeros by the row's workgroup)
            lo, hi = {None: (0, self.n_pack_rows), "first": (0, self.n_first_rows), "rest": (self.n_first_rows, self.n_pack_rows)}[part]
            if hi > lo:
                nat.call("ctseg_pack_weights", st.flat_p.data_ptr(), self.pack_blocks.data_ptr(), self.n_pack_blocks,
                         self.pack_rows.data_ptr() + 8 * lo, hi - lo, self.buf.data_ptr(), self.plan.dt)
        elif part != "rest":
            nat.call("ctseg_gather_cast", st.flat_p.data_ptr(), self.idx.data_ptr(), self.buf.data_ptr(), self.plan.dt, self.n_idx)
        if part != "rest":
            nat.call("ctseg_gather_cast", st.flat_p.data_ptr(), self.bias_idx.data_ptr(), self.bias_buf.data_ptr(), F32, self.n_bidx)
        if part != "first":
            self.dirty, self.version = False, ver
