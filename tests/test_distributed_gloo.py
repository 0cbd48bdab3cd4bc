"""CPU, world_size 2, gloo: the data-parallel path of capstone_amd.distributed — chunked all-reduce of the flat
gradient buffer with the readiness hooks Plan.backward fires, 1/world folded into the optimizer's grad_scale.
(On the GPU box the same code runs over RCCL/xGMI with backend "nccl".)"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from capstone_amd import distributed as cdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, _, w = cdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    n = 1000
    sizes = {0: 100, 100: 400, 500: 300, 800: 200}                      # four "parameters" in readiness order
    marks = [(3, [0]), (7, [100, 500]), (12, [800])]                      # (backward program index, offsets final)
    flat_g = torch.arange(n, dtype=torch.float32) * (rank + 1)
    flat_p = torch.full((n,), float(rank))
    cdist.broadcast_params(flat_p)
    red = cdist.GradAllReducer(flat_g, n, marks, sizes)
    hooks = red.hooks()
    fired = []
    for idx in range(15):                                                 # what Plan.backward does between ops
        if idx in hooks:
            hooks[idx]()
            fired.append((idx, red.sent))
    scale = red.finish()
    q.put((rank, fired, scale, flat_g.clone().numpy(), flat_p.numpy()))
    dist.destroy_process_group()


def test_flat_gradient_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    exp = np.arange(1000, dtype=np.float32) * 3.0                          # sum over ranks of (rank+1)*arange
    for rank, fired, scale, g, p in res:
        assert fired == [(7, 800)]          # first split point: earliest mark whose ready prefix covers >= 60 %
        assert scale == 0.5                 # mean over ranks is applied inside the Adam kernel
        np.testing.assert_array_equal(g, exp)
        np.testing.assert_array_equal(p, np.zeros(1000, dtype=np.float32))  # rank 0's parameters everywhere


def _worker_direct(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      CTSEG_DDP_ALGO="direct")
    cdist.init_from_env("gloo")
    n = 1003                                                              # chunks that do not divide by the world size
    sizes = {0: 101, 101: 400, 501: 300, 801: 202}
    marks = [(3, [0]), (7, [101, 501]), (12, [801])]
    g = torch.Generator().manual_seed(100 + rank)
    flat_g = torch.randn(n, generator=g)
    mine = flat_g.clone()
    red = cdist.GradAllReducer(flat_g, n, marks, sizes)
    assert red.algo == "direct"
    hooks = red.hooks()
    for idx in range(15):
        if idx in hooks:
            hooks[idx]()
    red.finish()
    q.put((rank, mine.numpy(), flat_g.clone().numpy()))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_direct_exchange_equals_the_sum_over_ranks(world):
    """CTSEG_DDP_ALGO=direct: all-to-all of the shards, local sum in rank order, all-gather — every rank must end with the same
    buffer, equal to the rank-ordered sum (the mean is the Adam kernel's grad_scale; Lightning DDP's gradient mean,
    capstone/volumetric/base_trainer.py:196,217).  Chunk lengths that do not divide by the world size exercise the all-reduced tail."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_direct, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    exp = res[0][1].copy()
    for _, mine, _ in res[1:]:
        exp = exp + mine                                                   # rank order, fp32
    for _, _, got in res:
        np.testing.assert_array_equal(got, res[0][2])                      # identical on every rank
        np.testing.assert_allclose(got, exp, rtol=0, atol=1e-6)


def test_split_points_follow_readiness_prefix():
    sizes = {0: 10, 10: 10, 20: 60, 80: 20}
    # offset 20 (the big bottleneck tensor) becomes final before offset 10: the prefix only advances when contiguous
    pts = cdist.split_points([(2, [0]), (5, [20]), (6, [10]), (9, [80])], sizes, 100)
    assert pts == [(6, 80)]
    assert cdist.split_points([(1, [0, 10, 20, 80])], sizes, 100) == []    # everything at once: one collective at the end
    # a second split where >= 92 % is final before the end of the backward: only the last few per cent stay for the exposed collective
    sizes20 = {5 * i: 5 for i in range(20)}
    assert cdist.split_points([(i, [5 * i]) for i in range(20)], sizes20, 100) == [(11, 60), (18, 95)]
    # both fractions reached by the same mark: one split there
    assert cdist.split_points([(3, [0, 10, 20]), (9, [80])], sizes, 100) == [(3, 80)]


def test_world1_is_a_no_op():
    g = torch.ones(8)
    red = cdist.GradAllReducer(g, 8)
    assert red.hooks() == {} and red.finish() == 1.0 and torch.equal(g, torch.ones(8))


# ------------------------------------------------------------------------------------------------------------------------
# the data-parallel STEP, checked as a step: two ranks run the product's real fit_step (host logic + C-ABI emulator on CPU
# memory, gloo all-reduce fired from the backward hooks) on their own half of a batch; the oracle steps once per iteration on
# the concatenated batch.  Lightning DDP semantics (capstone/volumetric/base_trainer.py:196): identical replicas, gradient mean.
# ------------------------------------------------------------------------------------------------------------------------
FILTERS = (4, 8, 16)
STEPS = 3


def _ddp_batches():
    g = torch.Generator().manual_seed(77)
    out = []
    for r in range(2):
        x = torch.randn(1, 1, 8, 8, 8, generator=g)
        m = (torch.rand(1, 9, 8, 8, 8, generator=g) < 0.12).to(torch.uint8)
        out.append((x, m, torch.ones(1, 9)))
    return out


def _ddp_worker(rank, world, port, q, warm_local_steps):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from abi_emulator import Emulator, patch_native
    from capstone_amd import _native as nat, plan as plan_mod
    from capstone_amd.volumetric.base_trainer import BaseUNet3D
    e = Emulator()
    patch_native(nat, e)
    plan_mod.Plan.run = staticmethod(lambda prog, stream, lo=0, hi=None: e.run(prog[lo:hi]))
    cdist.init_from_env("gloo")
    torch.manual_seed(100 + rank)                     # DIFFERENT initial weights per rank: attach() must make them rank 0's
    m = BaseUNet3D(filters=list(FILTERS), loss_fx=["CrossEntropy"], lr=0.01)
    batch = _ddp_batches()[rank]
    for _ in range(warm_local_steps):                 # rank-local optimizer steps before attach: moments and step count diverge
        m.fit_step(batch)
    m.unet.engine().ensure("cpu")
    red = cdist.attach(m)
    st = m.unet.engine().store
    p_attach = st.flat_p.clone().numpy()
    losses = []
    for _ in range(STEPS):
        losses.append(float(m.fit_step(batch)))
        assert red.points_for(m.unet.engine().last_plan), "a chunk must go out mid-backward"
    q.put((rank, p_attach, st.flat_p.clone().numpy(), st.adam_m.clone().numpy(), st.adam_v.clone().numpy(), st.step, losses,
           {k: v.clone().numpy() for k, v in m.state_dict().items()}))
    dist.destroy_process_group()


def _run_ddp(warm_local_steps):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q, warm_local_steps)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


def test_two_rank_fit_step_equals_one_process_on_the_concatenated_batch():
    from oracle.trainer import OracleUNet3D
    r0, r1 = _run_ddp(0)
    # (i) replicas: bit-identical weights, moments and step count on both ranks after K steps
    for i in (1, 2, 3, 4):
        np.testing.assert_array_equal(r0[i], r1[i])
    assert r0[5] == r1[5] == STEPS
    # (ii) == one process stepping on the concatenated batch (CE is a mean over all voxels and both ranks hold the same number
    # of voxels, so the mean over ranks of the rank gradients is the gradient of the concatenated loss)
    torch.manual_seed(100)                             # rank 0's initial weights
    om = OracleUNet3D(filters=FILTERS, loss_fx=("CrossEntropy",), lr=0.01)
    sd0 = {k: v.clone() for k, v in om.state_dict().items()}
    opt = om.configure_optimizers()
    b = _ddp_batches()
    cat = tuple(torch.cat([b[0][i], b[1][i]]) for i in range(3))
    solid = {k: torch.ones_like(p, dtype=torch.bool) for k, p in om.named_parameters()}
    olosses = []
    for _ in range(STEPS):
        olosses.append(float(om.fit_step(cat, opt)))
        for k, p in om.named_parameters():
            gmax = float(p.grad.abs().max())
            solid[k] &= p.grad.abs() > 1e-3 * max(gmax, 1e-9)
    # loss of the concatenated batch = mean of the two rank losses
    np.testing.assert_allclose((np.array(r0[6]) + np.array(r1[6])) / 2, olosses, rtol=2e-4)
    moved = 0
    for k, p in om.named_parameters():
        got, ref = r0[7][k], p.detach().numpy()
        ok = solid[k].numpy()
        if k.endswith(".bias") and "residual" not in k:   # bias before an InstanceNorm: analytically zero gradient, noise only
            ok = ok & False
        # where the gradient is above the noise floor at every step Adam moved the weight the same way on both sides
        np.testing.assert_allclose(got[ok], ref[ok], rtol=0, atol=4e-4, err_msg=k)
        moved += int((np.abs(ref - sd0[k].numpy())[ok] > 5e-3).sum())
        # everywhere else the difference is bounded by K steps of size lr
        np.testing.assert_allclose(got, ref, rtol=0, atol=STEPS * 0.01 * 2.05, err_msg=k)
    assert moved > 100, "the comparison must cover weights that actually moved"


def test_attach_after_rank_local_steps_replicates_adam_state_too():
    """attach() after each rank has already taken optimizer steps of its own (the round-1 bench did one): weights, BOTH Adam
    moment buffers and the step count must become rank 0's, or the replicas apply different updates to the same mean gradient."""
    r0, r1 = _run_ddp(2)
    np.testing.assert_array_equal(r0[1], r1[1])        # weights right after attach
    for i in (2, 3, 4):
        np.testing.assert_array_equal(r0[i], r1[i])    # weights and both moments after K more steps
    assert r0[5] == r1[5] == 2 + STEPS


# ------------------------------------------------------------------------------------------------------------------------
# Round 4: the OTHER routes into the backward program.  training_step with Dice + Focal (the paper's recipe,
# capstone/training/base_trainer.py:29) does not take the fused cross-entropy node: its gradients come out of plan._UNetFn,
# which must run the same exchange — "every loss.backward() leaves the mean gradient" is Lightning DDP's contract
# (capstone/volumetric/base_trainer.py:196,217).  Also: forward() + a custom loss, the 2-D module's foreign conv1x1 parameter,
# the per-rank Dice counts gathered at epoch end, and the stock-DDP tripwire.
# ------------------------------------------------------------------------------------------------------------------------
DROPIN_LOSSES = ["Dice", "Focal"]


def _dropin_worker(rank, world, port, q, mode):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from abi_emulator import Emulator, patch_native
    from capstone_amd import _native as nat, plan as plan_mod
    from capstone_amd.volumetric.base_trainer import BaseUNet3D
    e = Emulator()
    patch_native(nat, e)
    plan_mod.Plan.run = staticmethod(lambda prog, stream, lo=0, hi=None: e.run(prog[lo:hi]))
    cdist.init_from_env("gloo")
    torch.manual_seed(100 + rank)
    m = BaseUNet3D(filters=list(FILTERS), loss_fx=list(DROPIN_LOSSES), lr=0.01)
    batch = _ddp_batches()[rank]
    out = {"rank": rank}
    # the tripwire: torch's DDP probes the module it wraps for ``_ddp_params_and_buffers_to_ignore``
    try:
        torch.nn.parallel.DistributedDataParallel(m)
        out["ddp_wrap"] = "accepted"
    except nat.NativeError as err:
        out["ddp_wrap"] = "NativeError" if "attach" in str(err) else str(err)
    m.unet.engine().ensure("cpu")
    red = cdist.attach(m)
    assert m.reducer is red and m.unet.engine().reducer is red
    opt = m.configure_optimizers()
    st = m.unet.engine().store
    losses, fired = [], []
    for _ in range(STEPS):
        opt.zero_grad()
        if mode == "training_step":
            loss = m.training_step(batch, 0)
        else:                                   # forward() + the loss wrapper by hand: the same node, another caller
            from capstone_amd.volumetric.utils import _squash_masks_3D
            y = m(batch[0])
            y._ctseg_plan = m.unet.engine().last_plan
            d = m.loss_func(input=y, target=_squash_masks_3D(batch[1], 10, "cpu"), mask_indicator=batch[2])
            loss = torch.stack(list(d.values())).sum()
        loss.backward()
        fired.append(len(red.points_for(m.unet.engine().last_plan)))
        g0 = next(iter(m.parameters())).grad.clone().numpy()       # p.grad = the MEAN over ranks (scaled in publish())
        opt.step()
        losses.append(float(loss))
    dice = m.epoch_dice_across_ranks("train")
    local = m.epoch_means().get("Mean Dice Score (train)")
    out.update(p=st.flat_p.clone().numpy(), m=st.adam_m.clone().numpy(), v=st.adam_v.clone().numpy(), step=st.step, losses=losses,
               fired=fired, g0=g0, sd={k: v.clone().numpy() for k, v in m.state_dict().items()},
               dice=None if dice is None else (float(dice[0]), dice[1].numpy()),
               local_dice=None if local is None else float(local))
    q.put(out)
    dist.destroy_process_group()


def _run_dropin(mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dropin_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t["rank"])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("mode", ["training_step", "forward_plus_loss"])
def test_two_rank_dice_focal_training_step_backward_step_keeps_replicas_identical(mode):
    from oracle.trainer import OracleUNet3D
    r0, r1 = _run_dropin(mode)
    assert r0["ddp_wrap"] == r1["ddp_wrap"] == "NativeError"
    # (i) replicas bit-identical: weights, both moments, step count, and the p.grad autograd saw
    for k in ("p", "m", "v", "g0"):
        np.testing.assert_array_equal(r0[k], r1[k], err_msg=k)
    assert r0["step"] == r1["step"] == STEPS
    assert all(n >= 1 for n in r0["fired"]), "a chunk must go out mid-backward on this route too"
    # (ii) == the oracle stepping on the concatenated batch (Dice: mean over (sample, class); Focal: mean over voxels per
    # (sample, class), then mean — both are means over samples, so the rank mean of the rank gradients is the global gradient)
    torch.manual_seed(100)
    om = OracleUNet3D(filters=FILTERS, loss_fx=tuple(DROPIN_LOSSES), lr=0.01)
    opt = om.configure_optimizers()
    b = _ddp_batches()
    cat = tuple(torch.cat([b[0][i], b[1][i]]) for i in range(3))
    solid = {k: torch.ones_like(p, dtype=torch.bool) for k, p in om.named_parameters()}
    olosses, odice = [], []
    for _ in range(STEPS):
        olosses.append(float(om.fit_step(cat, opt)))
        odice.append(float(om.logged["Mean Dice Score (train)"]))
        for k, p in om.named_parameters():
            solid[k] &= p.grad.abs() > 1e-3 * max(float(p.grad.abs().max()), 1e-9)
    np.testing.assert_allclose((np.array(r0["losses"]) + np.array(r1["losses"])) / 2, olosses, rtol=5e-4)
    checked = 0
    for k, p in om.named_parameters():
        got, ref = r0["sd"][k], p.detach().numpy()
        ok = solid[k].numpy()
        if k.endswith(".bias") and "residual" not in k:
            ok = ok & False
        np.testing.assert_allclose(got[ok], ref[ok], rtol=0, atol=6e-4, err_msg=k)
        np.testing.assert_allclose(got, ref, rtol=0, atol=STEPS * 0.01 * 2.05, err_msg=k)
        checked += int(ok.sum())
    assert checked > 100
    # (iii) epoch Dice over the GLOBAL batch from the gathered integer counts == the oracle's Dice on the concatenated batch,
    # identical on both ranks; the rank-local epoch mean (what the reference logs per rank) is a different number in general
    if mode == "training_step":          # (forward() + a hand-made loss logs nothing: no counts were kept)
        assert r0["dice"] is not None and r0["dice"][0] == r1["dice"][0]
        np.testing.assert_array_equal(r0["dice"][1], r1["dice"][1])
        np.testing.assert_allclose(r0["dice"][0], np.mean(odice), atol=2e-3)


def _foreign_worker(rank, world, port, q):
    """BaseUNet2D(--downsample): conv1x1 is NOT in the flat store; attach() broadcasts it and averages its gradient by a hook"""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from abi_emulator import Emulator, patch_native
    from capstone_amd import _native as nat, plan as plan_mod
    from capstone_amd.training.base_trainer import BaseUNet2D
    e = Emulator()
    patch_native(nat, e)
    plan_mod.Plan.run = staticmethod(lambda prog, stream, lo=0, hi=None: e.run(prog[lo:hi]))
    cdist.init_from_env("gloo")
    torch.manual_seed(200 + rank)
    m = BaseUNet2D(filters=[4, 8, 16, 32, 64], downsample=True, lr=0.01, transform_degree=1)
    g = torch.Generator().manual_seed(300 + rank)
    x = torch.randn(1, 3, 32, 32, generator=g)
    mk = (torch.rand(1, 9, 32, 32, generator=g) < 0.1).to(torch.uint8)
    batch = (x, mk, torch.ones(1, 9))
    m.unet.engine().ensure("cpu")
    wrapper = m.configure_ddp(m, None)             # the Lightning hook, called by hand: attach + pass-through wrapper
    assert wrapper.module is m and m.reducer is not None
    opt = m.configure_optimizers()["optimizer"]
    m.train()
    for _ in range(2):
        opt.zero_grad()
        loss = wrapper(batch, 0)                   # dispatches to training_step, as LightningDistributedDataParallel.forward
        loss.backward()
        opt.step()
    q.put((rank, m.conv1x1.weight.detach().clone().numpy(), m.conv1x1.bias.detach().clone().numpy(),
           m.unet.engine().store.flat_p.clone().numpy()))
    dist.destroy_process_group()


def test_two_rank_2d_downsample_foreign_parameters_stay_replicated():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_foreign_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for i in (1, 2, 3):
        np.testing.assert_array_equal(res[0][i], res[1][i])
