#!/usr/bin/env python3
"""per-queue busy time and idle gaps of the last full step in a rocprofv3 kernel_trace.csv: tools/trace_gaps.py trace.csv"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "squash_masks" in r["Kernel_Name"]]
lo, hi = marks[-3], marks[-2]
t0, t1 = int(rows[lo]["Start_Timestamp"]), int(rows[hi]["Start_Timestamp"])
print("step span %.1f us, %d launches" % ((t1 - t0) / 1e3, hi - lo))
byq = collections.defaultdict(list)
for r in rows[lo:hi]:
    byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
for q, ks in byq.items():
    busy = sum(e - s for s, e, _ in ks)
    gaps = [(ks[i + 1][0] - ks[i][1], ks[i][2][:60], ks[i + 1][2][:60]) for i in range(len(ks) - 1)]
    pos = [g for g in gaps if g[0] > 0]
    print(f"queue {q}: {len(ks)} launches, busy {busy / 1e3:.1f} us, first {(ks[0][0] - t0) / 1e3:.1f}, last end {(ks[-1][1] - t0) / 1e3:.1f}, "
          f"sum of gaps {sum(g[0] for g in pos) / 1e3:.1f} us ({len(pos)} gaps, median {sorted(g[0] for g in pos)[len(pos) // 2] / 1e3:.2f} us)")
    for g in sorted(pos, reverse=True)[:8]:
        print(f"     gap {g[0] / 1e3:7.1f} us  after {g[1]} -> {g[2]}")
if len(sys.argv) > 2:
    for q, ks in byq.items():
        agg = collections.defaultdict(lambda: [0, 0])
        for s, e, n in ks:
            agg[n[:100]][0] += 1; agg[n[:100]][1] += e - s
        print(f"---- queue {q}")
        for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2])]:
            print(f"  {t / 1e3:8.1f} us  x{c:3d}  {n}")
