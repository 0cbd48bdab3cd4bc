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


def test_split_points_follow_readiness_prefix():
    sizes = {0: 10, 10: 10, 20: 60, 80: 20}
    # offset 20 (the big bottleneck tensor) becomes final before offset 10: the prefix only advances when contiguous
    pts = cdist.split_points([(2, [0]), (5, [20]), (6, [10]), (9, [80])], sizes, 100)
    assert pts == [(6, 80)]
    assert cdist.split_points([(1, [0, 10, 20, 80])], sizes, 100) == []    # everything at once: one collective at the end


def test_world1_is_a_no_op():
    g = torch.ones(8)
    red = cdist.GradAllReducer(g, 8)
    assert red.hooks() == {} and red.finish() == 1.0 and torch.equal(g, torch.ones(8))
