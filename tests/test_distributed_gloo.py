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
