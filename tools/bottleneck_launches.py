"""Per-dispatch durations of the encoder-bottleneck Conv3d (forward 256->256 k3) from a rocprofv3 --kernel-trace csv of bench.py:
conv_igemm_ring_kernel<BF16, 256> is launched five times per training step (forward 64->256 s2, 128->256, 256->256; input
gradients 256->256 and 256->128 as Cn=256 passes) — the third of every five is the bottleneck forward."""
import csv
import json
import sys


def main(trace, bench_json=None, steps=10):
    rows = [r for r in csv.DictReader(open(trace)) if "conv_igemm_ring_kernel<ctseg::BF16, 256>" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
    assert len(d) % 5 == 0, len(d)
    third = d[2::5][-steps:]
    avg = sum(third) / len(third)
    out = {"kernel": "conv_igemm_ring_kernel<BF16, 256> — 3rd of its 5 launches per step = encoder-bottleneck Conv3d 256->256 k3 forward",
           "source": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline "
                     "--fp32-steps 0 (kernel_trace.csv, per-dispatch end - start)",
           f"durations_ns_last{steps}_steps": third, "average_ns": avg, "tflops": 173946175488.0 / avg / 1e3,
           "frac_of_2500_tflops": 173946175488.0 / avg / 1e3 / 2500.0}
    if bench_json:
        b = json.load(open(bench_json))
        out["bench_py_same_run"] = {k: b["roofline"][k] for k in ("launch_ms", "launch_ms_minus_event_pair", "event_pair_ms", "frac")}
        out["bench_py_same_run"]["ms_per_step"] = b["ms_per_step"]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:3])
