#!/usr/bin/env python3
"""compare two rocprofv3 kernel_stats.csv files: tools/cmp_stats.py new.csv old.csv [top]"""
import csv, sys
def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        d[r['Name']] = (int(r['Calls']), float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6)
    return d
new, old = load(sys.argv[1]), load(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
steps = 25.0
print("kernel ms per step: new %.3f old %.3f" % (sum(v[2] for v in new.values()) / steps, sum(v[2] for v in old.values()) / steps))
keys = sorted(set(new) | set(old), key=lambda k: -max(new.get(k, (0, 0, 0))[2], old.get(k, (0, 0, 0))[2]))
for k in keys[:top]:
    n, o = new.get(k, (0, 0, 0)), old.get(k, (0, 0, 0))
    print(f"{k[:105]:105s} new {n[0]:4d} x {n[1]:7.1f} = {n[2]/steps*1e3:7.1f} us/step | old {o[0]:4d} x {o[1]:7.1f} = {o[2]/steps*1e3:7.1f}")
