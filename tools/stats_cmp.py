#!/usr/bin/env python3
"""per-kernel average duration of two rocprofv3 --stats csv files side by side:  stats_cmp.py A.csv B.csv [filter]"""
import csv
import sys


def load(p):
    return {r["Name"]: (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6)
            for r in csv.DictReader(open(p))}


a, b = load(sys.argv[1]), load(sys.argv[2])
flt = sys.argv[3] if len(sys.argv) > 3 else ""
ta = tb = 0.0
for k in sorted(set(a) | set(b), key=lambda k: -(b.get(k, (0, 0, 0))[2])):
    if flt and flt not in k:
        continue
    ca, ua, ma = a.get(k, (0, 0.0, 0.0))
    cb, ub, mb = b.get(k, (0, 0.0, 0.0))
    ta += ma
    tb += mb
    if max(ma, mb) > 1.0:
        print(f"{k[:64]:64s} {ca:5d} {ua:8.1f}us {ma:8.2f}ms | {cb:5d} {ub:8.1f}us {mb:8.2f}ms")
print(f"total {ta:.1f} ms | {tb:.1f} ms")
