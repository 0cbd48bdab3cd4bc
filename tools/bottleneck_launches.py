"""Per-dispatch durations of the encoder-bottleneck Conv3d (forward 256->256 k3) from a rocprofv3 --kernel-trace csv of bench.py:
conv_igemm_ring_kernel<BF16, 256, false> (the instantiation without the backward-statistics epilogue, round 3) is launched three
times per training step — forward 64->256 s2, 128->256, 256->256 — the third of every three is the bottleneck forward.  (Rounds 1-2:
conv_igemm_ring_kernel<BF16, 256>, five launches per step, the third of every five.)"""
import csv
import json
import sys


def main(trace, bench_json=None, steps=10):
    allrows = list(csv.DictReader(open(trace)))
    rows = [r for r in allrows if "conv_igemm_ring_kernel<ctseg::BF16, 256, false>" in r["Kernel_Name"]]
    per = 3
    if not rows:
        rows, per = [r for r in allrows if "conv_igemm_ring_kernel<ctseg::BF16, 256>" in r["Kernel_Name"]], 5
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
    assert len(d) % per == 0 and d, len(d)
    third = d[2::per][-steps:]
    avg = sum(third) / len(third)
    out = {"kernel": f"conv_igemm_ring_kernel<BF16, 256{', false' if per == 3 else ''}> — 3rd of its {per} launches per step = encoder-bottleneck Conv3d 256->256 k3 forward",
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
