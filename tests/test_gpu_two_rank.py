"""GPU: the data-parallel path with TWO processes on the real HIP kernels ("no N > 1 run exists").  The pool's boxes have one
MI355X and RCCL refuses two ranks on one device, so the two ranks share cuda:0 and the transport is gloo; what runs is otherwise the
N > 1 path as bench.py --gpus N runs it: init_process_group before the first GPU call, attach() (replicas from rank 0: weights, Adam
moments, step count), the chunked flat-gradient exchange fired by the readiness hooks of the recorded backward from the
weight-gradient stream, the mean folded into the Adam kernel (fit_step) or applied by publish() (loss.backward()).  Ranks start from
DIFFERENT weights and see different data; after three steps their weights and moments must be bit-identical (no float atomics anywhere,
gloo's all-reduce gives every rank the same sums)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("mode", ["fit_step_ce", "training_step_dice_focal"])
def test_two_processes_on_one_card_keep_bit_identical_replicas(mode):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "gpu_ddp_child.py"), str(r), "2", str(port), mode],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=600)
        assert p.returncode == 0, so[-2000:] + "\n" + se[-5000:]
        line = [ln for ln in so.splitlines() if ln.startswith("GPU_DDP_CHILD ")][-1]
        outs.append(json.loads(line[len("GPU_DDP_CHILD "):]))
    a, b = sorted(outs, key=lambda o: o["rank"])
    assert a["finite"] and b["finite"]
    assert a["weights_sha"] == b["weights_sha"] and a["moments_sha"] == b["moments_sha"], (a, b)
    assert a["step"] == b["step"] == 3
    assert a["chunks"] >= 2, "a chunk of the gradient must go out before the backward ends"
    assert a["losses"] != b["losses"]                      # different shards
    assert a["global_dice"] is not None and a["global_dice"] == b["global_dice"]
