"""Host-side cost of enqueueing one training step (the recorded program is replayed by a Python loop over ctypes calls): wall time of
K fit_step calls WITHOUT a device synchronize vs with one.  If the two are close the step is launch-bound on the host."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ct-image-segmentation_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench as B


def main():
    dev = torch.device("cuda:0")
    from capstone_amd.volumetric.base_trainer import BaseUNet3D
    torch.manual_seed(B.SEED)
    m = BaseUNet3D(filters=list(B.FILTERS), loss_fx=["CrossEntropy"], precision="bf16", batch_size=2).to(dev)
    batch = B.synthetic_batch(2, 512, 512, 48, dev, B.SEED)
    for _ in range(5):
        m.fit_step(batch, keep_logits=False)
    torch.cuda.synchronize()
    K = 30
    t0 = time.perf_counter()
    for _ in range(K):
        m.fit_step(batch, keep_logits=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"enqueue {1e3 * (t1 - t0) / K:.2f} ms/step on the host, {1e3 * (t2 - t0) / K:.2f} ms/step with the device")


if __name__ == "__main__" and len(sys.argv) == 1:
    main()


def breakdown():
    """python tools/time_host_enqueue.py breakdown: host time inside Plan.run (the ctypes replay) vs the rest of fit_step"""
    dev = torch.device("cuda:0")
    from capstone_amd.volumetric.base_trainer import BaseUNet3D
    from capstone_amd import plan as P
    torch.manual_seed(B.SEED)
    m = BaseUNet3D(filters=list(B.FILTERS), loss_fx=["CrossEntropy"], precision="bf16", batch_size=2).to(dev)
    batch = B.synthetic_batch(2, 512, 512, 48, dev, B.SEED)
    for _ in range(5):
        m.fit_step(batch, keep_logits=False)
    torch.cuda.synchronize()
    acc = {"run": 0.0, "ops": 0, "calls": 0}
    orig = P.Plan.run

    def timed(prog, stream, lo=0, hi=None):
        t = time.perf_counter()
        orig(prog, stream, lo, hi)
        acc["run"] += time.perf_counter() - t
        acc["ops"] += len(prog[lo:hi])
        acc["calls"] += 1
    P.Plan.run = staticmethod(timed)
    K = 30
    t0 = time.perf_counter()
    for _ in range(K):
        m.fit_step(batch, keep_logits=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"fit_step host {1e3 * (t1 - t0) / K:.2f} ms/step: Plan.run {1e3 * acc['run'] / K:.2f} ms in {acc['calls'] / K:.0f} calls, "
          f"{acc['ops'] / K:.0f} recorded ops ({1e6 * acc['run'] / max(acc['ops'], 1):.1f} us per op)")
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        m.fit_step(batch, keep_logits=False)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "breakdown":
    breakdown()
