#!/usr/bin/env python3
"""Per-op timing of the recorded forward/backward programs of the config-B plan (HIP events around each C-ABI call).

    python tools/bench_layers.py [--shape 2 512 512 48] [--precision bf16] [--reps 5] [--only IDX --loop N]

Prints one line per op: program, index, entry point, geometry, ms, achieved TFLOP/s (conv passes) and the GB/s the
op would need if it touched its operands exactly once (algorithmic bytes).  `--only fwd:IDX --loop N` replays a single
op N times (for rocprofv3 --pmc runs on one kernel)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ct-image-segmentation_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

from bench import FILTERS, SEED, synthetic_batch  # noqa: E402
from capstone_amd import _native as nat  # noqa: E402
from capstone_amd.volumetric.base_trainer import BaseUNet3D  # noqa: E402


def describe(name, args, dt):
    sz = 2 if dt == nat.BF16 else 4
    if name == "ctseg_conv_igemm":
        d = args[0]
        rows = d.N * d.Xr * d.Yr * d.Zr
        taps = sum(d.cls[i].ntaps for i in range(d.nclass))
        flop = 2.0 * rows * d.Cn * d.Cg * taps
        osz = 4 if d.out_f32 else sz
        out_b = d.N * d.Xo * d.Yo * d.Zo * d.Cn_store * osz
        byt = d.N * d.Xi * d.Yi * d.Zi * d.Cg * sz + out_b
        if d.add and d.add != d.in_:
            # an addend is a SECOND read stream only when it is another tensor, and then in ITS storage type (fp32 logits take a
            # 16-bit addend); an identity residual (add == in) is served from the staged operand: 0 algorithmic bytes
            byt += d.N * d.Xo * d.Yo * d.Zo * d.Cn_store * (4 if d.add_f32 else sz)
        return f"conv Cg={d.Cg:3d} Cn={d.Cn:3d} in={d.Xi}x{d.Yi}x{d.Zi} rows={d.Xr}x{d.Yr}x{d.Zr} cls={d.nclass} s={d.sin}/{d.sout}{' +bst' if d.bst_partials else ''}" \
               f"{' +add' if d.add else ''}{' +stats' if d.stats else ''}{' f32out' if d.out_f32 else ''}", flop, byt
    if name == "ctseg_conv_wgrad":
        d = args[0]
        rows = d.N * d.Xr * d.Yr * d.Zr
        flop = 2.0 * rows * d.Cn * d.Cg * d.ntaps
        byt = d.N * d.Xi * d.Yi * d.Zi * d.Cg * sz + rows * d.Cn * sz
        return f"wgrad Cg={d.Cg:3d} Cn={d.Cn:3d} in={d.Xi}x{d.Yi}x{d.Zi} rows={d.Xr}x{d.Yr}x{d.Zr} s={d.sin} splits={d.splits}", flop, byt
    if name.startswith("ctseg_instnorm_prelu_fwd"):
        N, S, C = args[9], args[10], args[11]
        return f"in+prelu fwd C={C} S={S}{' +res' if args[5] else ''}", 0.0, N * S * C * sz * (3 if args[5] else 2)
    if name.startswith("ctseg_instnorm_prelu_bwd_reduce"):
        N, S, C = args[10], args[11], args[12]
        return f"in+prelu bwd reduce C={C} S={S}", 0.0, N * S * C * sz * 2
    if name.startswith("ctseg_instnorm_prelu_bwd_apply"):
        N, S, C = args[12], args[13], args[14]
        return f"in+prelu bwd apply C={C} S={S}{' +gcopy' if args[10] else ''}", 0.0, N * S * C * sz * (4 if args[10] else 3)
    return "", 0.0, 0.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", type=int, nargs=4, default=[2, 512, 512, 48])
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default=None)
    ap.add_argument("--loop", type=int, default=20)
    ap.add_argument("--min-ms", type=float, default=0.03)
    ap.add_argument("--filters", type=int, nargs="+", default=list(FILTERS))
    ap.add_argument("--infer", action="store_true", help="inference-only plan (forward program of a sliding-window batch)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(SEED)
    m = BaseUNet3D(filters=a.filters, loss_fx=["CrossEntropy"], precision=a.precision).to(dev)
    if a.infer:
        plan = m.unet.engine().plan_for_shape(dev, a.shape[0], a.shape[1:], inference=True)
        plan.x.t.normal_()
        plan.forward()
        plan.forward()
    else:
        batch = synthetic_batch(*a.shape, dev, SEED)
        m.fit_step(batch)
        m.fit_step(batch)
        plan = m.unet.engine().last_plan
    st = nat.stream_ptr()
    progs = {"fwd": plan.fwd, "bwd": plan.bwd}
    if a.only:
        which, idx = a.only.split(":")
        name, fn, args = progs[which][int(idx)]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn(*args, st)
        e0.record()
        for _ in range(a.loop):
            fn(*args, st)
        e1.record()
        torch.cuda.synchronize()
        print("looped", which, idx, name, describe(name, args, plan.dt)[0], f"avg {e0.elapsed_time(e1) / a.loop:.4f} ms back-to-back")
        return
    total = 0.0
    for which, prog in progs.items():
        for i, (name, fn, args) in enumerate(prog):
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.reps)]
            for s, e in ev:
                s.record()
                fn(*args, st)
                e.record()
            torch.cuda.synchronize()
            ms = sorted(s.elapsed_time(e) for s, e in ev)[len(ev) // 2]
            total += ms
            desc, flop, byt = describe(name, args, plan.dt)
            if ms >= a.min_ms:
                gbs = byt / ms / 1e6 if byt else 0
                print(f"{which}:{i:3d} {name[6:]:28s} {ms:8.3f} ms  {flop / ms / 1e9 if flop else 0:7.1f} TF/s  "
                      f"{gbs:7.0f} GB/s(alg){' (> HBM peak: operands partly cache-resident)' if gbs > 8000 else ''}  {desc}")
    print(f"sum of per-op medians: {total:.2f} ms; recorded launches: forward {len(plan.fwd)}, backward {len(plan.bwd)}")


if __name__ == "__main__":
    main()
