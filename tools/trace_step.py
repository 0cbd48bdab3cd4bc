#!/usr/bin/env python3
"""per-dispatch durations of one steady-state step from a rocprofv3 kernel_trace.csv: tools/trace_step.py trace.csv [substr]
prints, for the LAST full step (between two squash_masks launches), every launch whose name contains substr, in start order"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "squash_masks" in r["Kernel_Name"]]
lo, hi = marks[-3], marks[-2]
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:hi]:
    if sub in r["Kernel_Name"]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  q{r.get('Queue_Id','?')}  {r['Kernel_Name'][:110]}")
print("step span", (int(rows[hi]["Start_Timestamp"]) - t0) / 1e3, "us")
