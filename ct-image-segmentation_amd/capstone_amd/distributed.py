"""Data-parallel training: one process per GPU, RCCL all-reduce of the flat gradient buffer over xGMI.

The reference has no distributed code of its own; multi-GPU is Lightning's DDP (gradient MEAN over
ranks, bucketed NCCL all-reduce overlapped with backward) reached through Trainer flags
(capstone/volumetric/base_trainer.py:196,217).  Here the whole gradient is one flat fp32 buffer laid
out in gradient-readiness order, so the exchange is three collectives per step, not one per tensor:
chunk 1 (decoder + bottleneck, ~76 % of the bytes) is launched as soon as the bottleneck's weight
gradient retires and overlaps the encoder backward; chunk 2 (the deeper encoder levels, up to >= 92 %) overlaps the
level-0 backward; only the last few per cent (level 0 + stem) go behind the backward.  The 1/world
factor is folded into the Adam kernel's ``grad_scale`` (no extra pass).  Samples are independent
(InstanceNorm is per sample), so there is no other data-path collective.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as set by ``python -m torch.distributed.run``."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return 0, 0, 1
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend is None:
        # "nccl" is RCCL on ROCm.  CTSEG_DIST_BACKEND=gloo + CTSEG_SINGLE_DEVICE=1 rehearse the N>1 path on a one-GPU box
        backend = os.environ.get("CTSEG_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if os.environ.get("CTSEG_SINGLE_DEVICE") == "1":
        local = 0
    if torch.cuda.is_available() and (backend != "gloo" or local < torch.cuda.device_count()):
        torch.cuda.set_device(local)       # a gloo run on CPU tensors may have more ranks than the box has GPUs
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


SPLIT_FRACTIONS = (0.6, 0.92)      # share of the flat gradient that is final when chunk 1 / chunk 2 of the exchange goes out


def split_points(ready_marks, sizes, n_total, fractions=SPLIT_FRACTIONS):
    """ready_marks: [(program index, [flat offsets that became final])], sizes: {offset: numel}.
    Returns [(program index, end offset)] — after ``program index`` ops of backward the flat gradient is
    final on [0, end).  For every fraction, the earliest mark whose ready prefix covers that share of the buffer (and
    extends the previous split): the first collective overlaps the encoder backward, the second leaves only the last few
    per cent (level 0 + stem) for the exposed collective behind the backward."""
    done, prefix, out = set(), 0, []
    order = sorted(sizes)
    pos = 0
    fi = 0
    for idx, offs in ready_marks:
        done.update(offs)
        while pos < len(order) and order[pos] in done:
            prefix = order[pos] + sizes[order[pos]]
            pos += 1
        while fi < len(fractions) and prefix >= fractions[fi] * n_total:
            if prefix < n_total and (not out or prefix > out[-1][1]):
                if out and out[-1][0] == idx:
                    out[-1] = (idx, prefix)
                else:
                    out.append((idx, prefix))
            fi += 1
    return out


class _EventWork:
    """a Work-like handle over a CUDA event: wait() makes the current stream wait for the communication stream"""

    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)


class GradAllReducer:
    def __init__(self, flat_g, n, ready_marks=None, sizes=None, group=None, always=False):
        self.flat_g, self.n, self.group = flat_g, n, group
        self.sizes = sizes
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # always=True issues the collectives on a one-rank group too (a one-GPU box can then drive RCCL's stream
        # hand-offs against the two-stream backward; the sums are the identity there)
        self.always = bool(always)
        self.active = self.world > 1 or (always and dist.is_initialized())
        self.points = split_points(ready_marks, sizes, n) if ready_marks else []
        self.works, self.sent = [], 0
        self.algo = os.environ.get("CTSEG_DDP_ALGO", "allreduce")      # "direct": all-to-all + local sum + all-gather
        self._keep = []

    def _launch(self, lo, hi):
        if hi <= lo or not self.active:
            return
        if self.algo == "direct" and (self.world > 1 or self.always) and hi - lo >= self.world:
            self._launch_direct(lo, hi)
        else:
            self.works.append(dist.all_reduce(self.flat_g[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _launch_direct(self, lo, hi):
        """CTSEG_DDP_ALGO=direct — the exchange SURVEY.md §5 prefers on a fully connected xGMI node, two hops instead of a ring's
        2 (W - 1): every rank sends shard k of the chunk straight to rank k (all-to-all = point-to-point copies over the direct
        links), sums the W copies of its own shard in rank order (deterministic), and the reduced shards travel back the same
        way (all-gather).  Issued from the stream the hook runs on; the tail that does not divide by W is all-reduced.
        Not the default: unmeasured on an 8-GPU node."""
        W = self.world
        shard = (hi - lo) // W
        body = self.flat_g[lo:lo + W * shard]
        if body.is_cuda:
            # the three legs run on a communication stream of their own, behind the stream the hook fires on: the
            # weight gradients queued behind the hook on that stream are not held up by the first leg
            if getattr(self, "_comm", None) is None:
                self._comm = torch.cuda.Stream(device=body.device)
            cur = torch.cuda.current_stream(body.device)
            self._comm.wait_stream(cur)
            with torch.cuda.stream(self._comm):
                recv = torch.empty_like(body)
                dist.all_to_all_single(recv, body, group=self.group)      # recv[k] = rank k's copy of MY shard
                mine = self._ordered_sum(recv.view(W, shard))
                dist.all_gather_into_tensor(body, mine, group=self.group)
                ev = self._comm.record_event()
            for t in (recv, mine, body):
                t.record_stream(self._comm)
            self.works.append(_EventWork(ev))
        else:
            recv = torch.empty_like(body)
            dist.all_to_all_single(recv, body, group=self.group)          # recv[k] = rank k's copy of MY shard
            mine = self._ordered_sum(recv.view(W, shard))
            self.works.append(dist.all_gather_into_tensor(body, mine, group=self.group, async_op=True))
        if lo + W * shard < hi:
            self.works.append(dist.all_reduce(self.flat_g[lo + W * shard:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self._keep.append((recv, mine))       # alive until finish(): the collectives read them asynchronously

    @staticmethod
    def _ordered_sum(parts):
        """sum over dim 0 in rank order, by explicit adds (not torch.sum, whose reduction order is an implementation detail):
        every rank computes bit-identical shards"""
        acc = parts[0].clone()
        for k in range(1, parts.shape[0]):
            acc += parts[k]
        return acc

    def points_for(self, plan):
        """split points of THIS plan's backward program (another batch shape, or the 16-wide fallback of the narrow-row layout,
        records a different program: its readiness marks sit at other indices), cached on the plan"""
        if plan is None or self.sizes is None:
            return self.points
        pts = getattr(plan, "_ddp_points", None)
        if pts is None:
            pts = plan._ddp_points = split_points(plan.ready_marks, self.sizes, self.n) if plan.ready_marks else []
        return pts

    def hooks(self, plan=None):
        """{backward program index: callable} — fired by Plan.backward between ops"""
        self.works, self.sent, self._keep = [], 0, []
        h = {}
        for idx, end in self.points_for(plan):
            def fire(end=end):
                self._launch(self.sent, end)
                self.sent = end
            h[idx] = fire
        return h

    def finish(self):
        """all-reduce the rest, make the compute stream wait for every chunk; returns the Adam grad_scale"""
        self._launch(self.sent, self.n)
        self.sent = self.n
        for w in self.works:
            w.wait()
        self.works = []
        self._keep = []
        return 1.0 / self.world


def broadcast_params(flat_p, group=None):
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_p, src=0, group=group)


def attach(module, group=None, always=False):
    """make a BaseUNet3D data-parallel: identical replicas on every rank + overlapped gradient all-reduce.

    A replica is the weights AND the optimizer state ("identical replicated weights + Adam state"; Lightning's
    DDP gets there by constructing every rank from the same seed before any step): rank 0's flat parameter buffer, both Adam
    moment buffers and the step count are broadcast, so attach() may be called at any point — before the first step (the
    moments are then zeros everywhere) or after rank-local warm-up steps — and the replicas leave it bit-identical.
    Needs ``engine.ensure(device)`` (or any forward) first so the flat buffers exist."""
    eng = module.unet.engine()
    st = eng.store
    assert st is not None, "run engine.ensure(device) (or one forward) before attach()"
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        st.ensure_adam_state()
        step = torch.tensor([st.step], dtype=torch.int64, device=st.flat_p.device)
        for t in (st.flat_p, st.adam_m, st.adam_v, step):
            dist.broadcast(t, src=0, group=group)
        st.step = int(step.item())
    st.touch()            # every plan's packed operands are rebuilt from the broadcast weights
    plan = eng.last_plan
    sizes = {st.off(p): p.numel() for p in st.params}
    # the reducer lives on the ENGINE: every route into the recorded backward program finds it there — fit_step, the fused
    # training_step's autograd node and the plain logits node (plan._UNetFn: Dice / Focal / GDL recipes, forward() + a custom loss)
    red = eng.reducer = GradAllReducer(st.flat_g, st.n, plan.ready_marks if plan is not None else None, sizes, group, always)
    _attach_foreign_parameters(module, st, group, red)
    return red


def _attach_foreign_parameters(module, store, group, red):
    """parameters of the module that are NOT views of the engine's flat buffer (BaseUNet2D.conv1x1 under ``--downsample``,
    capstone/training/base_trainer.py:53,81-85): rank 0's values are broadcast and their gradients are averaged where autograd
    produces them (a tensor hook: two tiny tensors, a blocking all-reduce each)."""
    own = {id(p) for p in store.params}
    foreign = [p for p in module.parameters() if id(p) not in own]
    if not foreign or not red.active or red.world < 2:
        return
    for p in foreign:
        dist.broadcast(p.data, src=0, group=group)
        if p.requires_grad and not getattr(p, "_ctseg_dp_hook", False):
            def mean(g, world=red.world, group=group):
                g = g.contiguous().clone()
                dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)
                return g / world
            p.register_hook(mean)
            p._ctseg_dp_hook = True


class NativeDataParallel(torch.nn.Module):
    """What ``configure_ddp`` hands Lightning INSTEAD of a ``DistributedDataParallel`` wrap (Lightning 1.0:
    ``LightningModule.configure_ddp(model, device_ids)`` returns the object the trainer then calls per batch; the reference reaches
    it through ``Trainer.from_argparse_args(args)`` with ``--gpus N --distributed_backend ddp``,
    capstone/volumetric/base_trainer.py:196,217).  A pass-through: the gradient mean is the engine's own flat-buffer exchange
    (``attach``), fired from inside the recorded backward program, so there is nothing for torch's bucketing reducer to do — and
    nothing it COULD do: the HIP kernels write gradients into the flat buffer, no AccumulateGrad node ever fires.  ``forward``
    dispatches like Lightning 1.0's ``LightningDistributedDataParallel.forward``: training_step / test_step / validation_step."""

    def __init__(self, module, device_ids=None):
        super().__init__()
        self.module = module
        self.device_ids = list(device_ids) if device_ids else []

    @staticmethod
    def _to(obj, device):
        if torch.is_tensor(obj):
            return obj if obj.device == device else obj.to(device, non_blocking=True)
        if isinstance(obj, (list, tuple)):
            return type(obj)(NativeDataParallel._to(o, device) for o in obj)
        if isinstance(obj, dict):
            return {k: NativeDataParallel._to(v, device) for k, v in obj.items()}
        return obj

    def forward(self, *inputs, **kwargs):
        m = self.module
        dev = next(m.parameters()).device           # DDP scatters the batch onto its one device first
        inputs, kwargs = self._to(inputs, dev), self._to(kwargs, dev)
        if m.training:
            return m.training_step(*inputs, **kwargs)
        if getattr(m, "testing", False) and hasattr(m, "test_step"):
            return m.test_step(*inputs, **kwargs)
        return m.validation_step(*inputs, **kwargs)


def gather_dice_counts(counts, group=None):
    """the integer Dice counts (|pred ∩ true|, |pred|, |true| per sample and class: 27 int64 per sample without
    the background) of every rank, concatenated along the sample axis in rank order.  counts: (..., B, 3, C) int64 — leading axes
    (the steps of an epoch) are kept, so ONE small all-gather at epoch end carries the whole epoch.  World size 1: returned as is."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) < 2:
        return counts
    world = dist.get_world_size(group)
    counts = counts.contiguous()
    parts = [torch.empty_like(counts) for _ in range(world)]
    dist.all_gather(parts, counts, group=group)
    return torch.cat(parts, dim=-3)
