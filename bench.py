#!/usr/bin/env python3
"""bench.py — CT volumes/sec of the 3-D U-Net training step (BASELINE.json's metric) on N MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one synthetic batch already resident in HBM: squash masks -> UNet forward
-> fused CE + Dice counts -> backward (dgrad + wgrad) -> [RCCL all-reduce of the flat gradient] -> Adam.
Workload = BASELINE.json configs[2]: MONAI-shaped UNet(3,1,10,(32,64,128,256),(2,2,2,2),num_res_units=2),
2 x 1 x 512 x 512 x 48 per GPU, bf16 storage / fp32 accumulate (weak scaling: per-GPU batch fixed).
Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` (encoder-bottleneck Conv3d,
timed with HIP events around its launches inside the timed steps) and `cpu_baseline` (the oracle step on the
host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "ct-image-segmentation_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

FILTERS = [32, 64, 128, 256]
SEED = 12342
PEAK_BF16_TFLOPS = 2500.0          # dense MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3


def synthetic_batch(B, H, W, D, device, seed):
    """randn images; 9 non-overlapping axis-aligned ellipsoids (~1.3 % foreground); all annotated."""
    g = torch.Generator(device=device).manual_seed(seed)
    images = torch.randn(B, 1, H, W, D, device=device, generator=g)
    masks = torch.zeros(B, 9, H, W, D, dtype=torch.uint8, device=device)
    xs = torch.arange(H, device=device).view(H, 1, 1).float()
    ys = torch.arange(W, device=device).view(1, W, 1).float()
    zs = torch.arange(D, device=device).view(1, 1, D).float()
    rel = [0.007, 0.3296, 0.0046, 0.2619, 0.3035, 0.0068, 0.0065, 0.0374, 0.0426]   # sizes ~ 1/WEIGHT
    vol = torch.tensor([1.0 / r for r in rel])
    vol = vol / vol.sum() * 0.0134 * H * W * D
    for c in range(9):
        cx, cy, cz = H * (0.15 + 0.08 * c), W * (0.2 + 0.07 * ((c * 5) % 9)), D * 0.5
        r = (float(vol[c]) * 3 / (4 * 3.14159)) ** (1 / 3)
        rz = min(r, D * 0.4)
        rxy = (float(vol[c]) * 3 / (4 * 3.14159 * rz)) ** 0.5
        m = ((xs - cx) / rxy) ** 2 + ((ys - cy) / rxy) ** 2 + ((zs - cz) / rz) ** 2 <= 1.0
        masks[:, c] = m.to(torch.uint8)
    # make them non-overlapping: a later class never claims a voxel an earlier one has
    taken = torch.zeros(B, H, W, D, dtype=torch.bool, device=device)
    for c in range(9):
        masks[:, c] &= (~taken).to(torch.uint8)
        taken |= masks[:, c].bool()
    return images, masks, torch.ones(B, 9, device=device)


def find_op(plan, name_sub, cin, cout):
    """index of the conv pass named like `name_sub` in the recorded forward program"""
    import capstone_amd._native as nat
    for i, (name, _, args) in enumerate(plan.fwd):
        if name == "ctseg_conv_igemm":
            d = args[0]
            if d.Cg == cin and d.Cn == cout and d.nclass == 1 and d.cls[0].ntaps == 27 and d.sin == 1:
                return i
    return None


def cpu_baseline(shape, threads, device=None, precision="bf16"):
    """the oracle (torch CPU restatement of the reference step) on the host cores: kind = "port".  The same step — same seeded
    weights, same seeded volume — then runs once on the device in the benchmark's storage dtype (and in fp32): BASELINE.json's
    "Dice vs ref" (second return value; the oracle computes its Dice inside the timed step anyway)."""
    from oracle.trainer import OracleUNet3D
    torch.set_num_threads(threads)
    torch.manual_seed(SEED)
    m = OracleUNet3D(filters=FILTERS, loss_fx=("CrossEntropy",))
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    opt = m.configure_optimizers()
    B, H, W, D = shape
    batch = synthetic_batch(B, H, W, D, "cpu", SEED)
    t0 = time.time()
    oloss = m.fit_step(batch, opt)
    dt = time.time() - t0
    frac = B * H * W * D / float(512 * 512 * 48)     # 512x512x48 volumes' worth of voxels in the sample
    base = {"value": frac / dt, "unit": "volumes/s", "cores": threads, "kind": "port",
            "sample": f"1 training step (fwd+loss+Dice+bwd+Adam) of the oracle on {B}x1x{H}x{W}x{D} fp32 "
                      f"(= {frac:.3f} of a 512x512x48 volume), {dt:.1f} s on {threads} threads"}
    dice = None
    if device is not None:
        from capstone_amd.volumetric.base_trainer import BaseUNet3D
        odice = float(m.logged["Mean Dice Score (train)"])
        oloss = float(m.logged["CrossEntropy Loss (train)"])
        dice = {"shape": [B, 1, H, W, D], "seed": SEED, "oracle_dice": odice, "oracle_loss": oloss,
                "what": "mean Dice (9 structures, NaN-aware batch mean) and cross-entropy of ONE training step from the same seeded "
                        "weights on the same seeded synthetic volume: torch-CPU oracle (fp32) vs the HIP step; north_star bar +-0.002"}
        dbatch = tuple(t.to(device) for t in batch)
        for prec in dict.fromkeys((precision, "fp32")):
            g = BaseUNet3D(filters=list(FILTERS), loss_fx=["CrossEntropy"], precision=prec, batch_size=B)
            g.load_state_dict(sd)
            g.to(device)
            gl = float(g.fit_step(dbatch, keep_logits=False))
            gd = float(g.logged["Mean Dice Score (train)"])
            key = "" if prec == precision else "_fp32"
            dice["dice" + key], dice["abs_diff" + key], dice["loss" + key] = gd, abs(gd - odice), gl
            dice["dtype" + key] = prec
            del g
        torch.cuda.empty_cache()
    return base, dice


def host_enqueue(model, batch, steps=10):
    """host side of one step: the recorded program is replayed by a Python loop over ctypes calls (no hipGraph).  Wall time of
    ``steps`` fit_step calls WITHOUT a device synchronize (the queue holds them: 10 steps = ~1400 launches) against the same with
    one — if the first approaches the second the step is launch-bound on the host."""
    plan = model.unet.engine().last_plan
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        model.fit_step(batch, keep_logits=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    n_launch = len(plan.fwd) + len(plan.bwd)
    return {"enqueue_ms_per_step": (t1 - t0) / steps * 1e3, "device_ms_per_step": (t2 - t0) / steps * 1e3,
            "recorded_ops_per_step": n_launch, "steps": steps,
            "note": "recorded C-ABI calls of the forward + backward programs; loss / Adam / re-layout / bookkeeping add ~10 more launches"}


def fp32_line(model, batch, steps):
    """the same workload in fp32 storage (v_mfma_f32_16x16x4_f32: the reference's own arithmetic), a few steps, N=1"""
    from capstone_amd.volumetric.base_trainer import BaseUNet3D
    dev = batch[0].device
    # (the bf16 model's buffers stay allocated: both plans together are a fraction of the 288 GB)
    torch.manual_seed(SEED)
    m = BaseUNet3D(filters=list(FILTERS), loss_fx=["CrossEntropy"], precision="fp32", batch_size=batch[0].shape[0]).to(dev)
    for _ in range(2):
        m.fit_step(batch, keep_logits=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = m.fit_step(batch, keep_logits=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"dtype": "f32", "steps": steps, "ms_per_step": dt * 1e3, "volumes_per_s": batch[0].shape[0] / dt,
            "loss_last_step": float(loss.item())}


def _time_steps(fn, steps, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def drop_in_line(batch, steps, precision):
    """The surface north_star names, driven the way Lightning 1.0's loop drives the reference (capstone/volumetric/base_trainer.py
    :80-82,113-114): ``training_step -> loss.backward() -> configure_optimizers().step() -> zero_grad()`` on a fresh module of
    the same workload, after the main measurement, N=1."""
    from capstone_amd.volumetric.base_trainer import BaseUNet3D
    dev = batch[0].device
    torch.manual_seed(SEED)
    m = BaseUNet3D(filters=list(FILTERS), loss_fx=["CrossEntropy"], precision=precision, batch_size=batch[0].shape[0]).to(dev)
    opt = m.configure_optimizers()
    last = {}

    def step():
        loss = m.training_step(batch, 0)
        loss.backward()
        opt.step()
        opt.zero_grad()
        last["loss"] = loss

    ms = _time_steps(step, steps)
    return {"surface": "training_step -> loss.backward() -> torch.optim.Adam.step() -> zero_grad()", "steps": steps,
            "ms_per_step": ms, "volumes_per_s": batch[0].shape[0] / (ms * 1e-3), "loss_last_step": float(last["loss"].item()),
            "optimizer": type(opt).__module__ + "." + type(opt).__name__}


def exchange_rehearsal(batch, steps, precision):
    """Fixed cost of the data-parallel exchange measurable on ONE card: a one-rank RCCL ("nccl") group, the reducer attached with
    always=True so every collective of the N>1 step is issued (sums are the identity), for both exchange algorithms, against the
    same module with the exchange off.  Reported under config.dp."""
    import torch.distributed as dist
    from capstone_amd import distributed as cdist
    from capstone_amd.volumetric.base_trainer import BaseUNet3D
    dev = batch[0].device
    made = False
    if not dist.is_initialized():
        import socket
        with socket.socket() as sk:              # a free port for the one-rank rendezvous
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
        made = True
    out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "steps": steps}
    torch.manual_seed(SEED)
    m = BaseUNet3D(filters=list(FILTERS), loss_fx=["CrossEntropy"], precision=precision, batch_size=batch[0].shape[0]).to(dev)
    m.fit_step(batch, keep_logits=False)
    out["ms_per_step_exchange_off"] = _time_steps(lambda: m.fit_step(batch, keep_logits=False), steps)
    prev = os.environ.get("CTSEG_DDP_ALGO")
    for algo in ("allreduce", "direct"):
        os.environ["CTSEG_DDP_ALGO"] = algo
        red = cdist.attach(m, always=True)
        ms = _time_steps(lambda: m.fit_step(batch, keep_logits=False), steps)
        pts = red.points_for(m.unet.engine().last_plan)
        out[f"ms_per_step_exchange_on[{algo}]"] = ms
        out["chunks"] = len(pts) + 1
        out["chunk_ends"] = [int(e) for _, e in pts] + [int(red.n)]
        m.reducer = None
    if prev is None:
        os.environ.pop("CTSEG_DDP_ALGO", None)
    else:
        os.environ["CTSEG_DDP_ALGO"] = prev
    out["ms_per_step_exchange_off_after"] = _time_steps(lambda: m.fit_step(batch, keep_logits=False), steps)
    if made:
        dist.destroy_process_group()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)     # ~2.4 s of timed GPU work at 12 ms/step: visible to a busy sampler
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--shape", type=int, nargs=4, default=[2, 512, 512, 48], metavar=("B", "H", "W", "D"))
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fp32-steps", type=int, default=5,
                    help="N=1 only: also time this many steps of the same workload in fp32 storage (the reference's own "
                         "arithmetic) after the main measurement; reported under config.fp32 (0 = skip)")
    ap.add_argument("--drop-in-steps", type=int, default=20,
                    help="N=1 only: also time this many steps of training_step -> loss.backward() -> Adam.step() (the reference's "
                         "own surface) after the main measurement; reported under config.drop_in (0 = skip)")
    ap.add_argument("--exchange-rehearsal", action="store_true",
                    help="N=1 only: time the step with the data-parallel exchange issued on a one-rank RCCL group "
                         "(both CTSEG_DDP_ALGO values) against the exchange off; reported under config.dp")
    ap.add_argument("--cpu-shape", type=int, nargs=4, default=[1, 512, 512, 48])   # one volume = half a batch: ~10 s on 16 host threads
    args = ap.parse_args()

    from capstone_amd import distributed as cdist
    from capstone_amd.volumetric.base_trainer import BaseUNet3D

    rank, local, world = cdist.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    B, H, W, D = args.shape

    torch.manual_seed(SEED)
    model = BaseUNet3D(filters=list(FILTERS), loss_fx=["CrossEntropy"], precision=args.precision, batch_size=B).to(dev)
    batch = synthetic_batch(B, H, W, D, dev, SEED + rank)
    if world > 1:
        # replicas BEFORE the first optimizer step: rank 0's weights (and the still-empty Adam state) everywhere, then every
        # step applies the same update to the same mean gradient
        model.unet.engine().ensure(dev)
        cdist.attach(model)
    # keep_logits=False: the logits convolution runs with the cross-entropy fused into its epilogue (the fp32 logits are an
    # intermediate of the training step; training_step / validation_step materialise them as the reference does)
    for _ in range(max(args.warmup, 1)):      # the first one builds the plan
        model.fit_step(batch, keep_logits=False)

    eng = model.unet.engine()
    plan = eng.last_plan
    probe_i = find_op(plan, "bottleneck", 256, 256)
    ev = []
    if probe_i is not None:
        orig_run = plan.run

        def run_probed(prog, stream, lo=0, hi=None):
            end = len(prog) if hi is None else hi       # (the forward may arrive in pieces: [0, 1) and [1, end) behind the split repack)
            if prog is plan.fwd and lo <= probe_i < end:
                a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                orig_run(prog, stream, lo, probe_i)
                a.record()
                orig_run(prog, stream, probe_i, probe_i + 1)
                b.record()
                c.record()          # empty bracket right behind: what a pair of event records costs by itself on this stream
                orig_run(prog, stream, probe_i + 1, hi)      # hi: the fused head stops the recorded forward before its last op
                ev.append((a, b, c))
            else:
                orig_run(prog, stream, lo, hi)
        plan.run = run_probed

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = model.fit_step(batch, keep_logits=False)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_v = float(loss.item())
    if probe_i is not None:
        plan.run = orig_run          # the event brackets belong to the timed region only

    baseline_shape = (B, H, W, D) == (2, 512, 512, 48)
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        out = {"metric": f"CT volumes/sec training ({H}x{W}x{D}, bs={B}/GPU)", "value": world * B * args.steps / elapsed,
               "unit": "volumes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
               "config": {"workload": f"3D U-Net (MONAI UNet, channels 32-64-128-256, 2 res units) training step on "
                                      f"{B}x1x{H}x{W}x{D} synthetic CT volumes per GPU ("
                                      + (f"BASELINE.json configs[2]{'' if world == 1 else '/[3]'}" if baseline_shape else
                                         "NOT the BASELINE.json shape 2x512x512x48")
                                      + "), CrossEntropy + Dice metric + Adam",
                          "global_batch": world * B, "parallelism": f"dp{world}", "loss_last_step": loss_v}}
        if world > 1:
            # what a driver needs to verify that RCCL saw N ranks and which exchange ran
            import torch.distributed as dist
            red = model.reducer
            out["config"]["dp"] = {"backend": dist.get_backend(), "world": dist.get_world_size(), "algo": red.algo,
                                   "chunks": len(red.points_for(plan)) + 1, "gradient_bytes": 4 * int(red.n)}
        if ev:
            raw = sum(a.elapsed_time(b) for a, b, _ in ev) / len(ev)
            ovh = sum(b.elapsed_time(c) for _, b, c in ev) / len(ev)
            # launch duration = the bracketed interval as measured: that is the figure the rocprofv3 per-dispatch average of the
            # same command agrees with (r01: 148.1 us vs 148.5 us bracket).  The interval of an empty bracket recorded right
            # behind it (what a pair of event records costs by itself) is reported beside it, not subtracted.
            kms = raw
            n_, x_, y_, z_ = B, H // 8, W // 8, D // 8
            flop = 2.0 * n_ * x_ * y_ * z_ * 256 * 256 * 27
            peak = PEAK_BF16_TFLOPS if args.precision == "bf16" else PEAK_F32_TFLOPS
            ach = flop / (kms * 1e-3) / 1e12
            # HBM bytes per launch are NOT measured in this run (PMC counters need rocprofv3): "traffic" stays null here and the
            # figure of the committed rocprofv3 --pmc passes of this very kernel is quoted under its own name, with its source
            traffic_committed, pmc_src = None, None
            pmc = sorted(p for p in os.listdir(os.path.join(ROOT, "profiles")) if p.endswith("pmc_bottleneck.json"))
            pmc = os.path.join(ROOT, "profiles", pmc[-1]) if pmc else ""
            if args.precision == "bf16" and (B, H, W, D) == (2, 512, 512, 48) and os.path.exists(pmc):
                traffic_committed, pmc_src = json.load(open(pmc))["traffic_bytes_per_launch"], os.path.relpath(pmc, ROOT)
            out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                               "traffic": None, "traffic_from_committed_pmc_profile": traffic_committed,
                               "traffic_source": pmc_src, "kernel": "conv_igemm_ring_kernel (192x256 tile, bf16) encoder-bottleneck Conv3d 256->256 k3",
                               "launch_ms": kms, "launch_ms_minus_event_pair": raw - ovh, "event_pair_ms": ovh,
                               "flop_per_launch": flop, "launches_timed": len(ev)}
            if baseline_shape:
                # the whole step against SURVEY.md §8(d)'s per-layer roofline: sum over the conv layers of
                # max(FLOP / 2.5 PFLOP/s, bytes / 6.29 TB/s), forward 0.89 ms, training step ~2.7 ms
                step_flop, step_roof_ms = 4092e9, 2.7
                out["roofline"]["step"] = {"flop": step_flop, "ms": ms, "tflops": step_flop / (ms * 1e-3) / 1e12,
                                           "frac_of_mfma_peak": step_flop / (ms * 1e-3) / 1e12 / peak,
                                           "per_layer_roofline_ms": step_roof_ms, "frac_of_per_layer_roofline": step_roof_ms / ms}
        if world == 1 and not args.no_cpu_baseline:
            try:
                ncpu = len(os.sched_getaffinity(0))
            except AttributeError:
                ncpu = os.cpu_count() or 1
            out["cpu_baseline"], dvr = cpu_baseline(tuple(args.cpu_shape), max(1, min(ncpu, 16)), dev, args.precision)
            out["config"]["dice_vs_ref"] = dvr
        if world == 1:
            out["config"]["host"] = host_enqueue(model, batch)
        if world == 1 and args.fp32_steps > 0 and args.precision == "bf16":
            out["config"]["fp32"] = fp32_line(model, batch, args.fp32_steps)
        if world == 1 and args.drop_in_steps > 0:
            out["config"]["drop_in"] = drop_in_line(batch, args.drop_in_steps, args.precision)
            out["config"]["drop_in"]["fit_step_ms_per_step"] = ms
        if world == 1 and args.exchange_rehearsal:
            out["config"]["dp"] = exchange_rehearsal(batch, max(10, min(args.steps, 50)), args.precision)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
