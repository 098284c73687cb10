#!/usr/bin/env python3
"""achieved HBM GB/s per kernel = PMC traffic per launch (profiles/rNN_pmc_traffic.json) / average duration in the
rocprofv3 --stats summary of the same workload (profiles/rNN_bench_kernel_stats_final.csv):
    python tools/hbm_summary.py profiles/r01_pmc_traffic.json profiles/r01_bench_kernel_stats_final.csv -o profiles/r01_hbm_gbps.json"""
import argparse
import csv
import json
import re

ap = argparse.ArgumentParser()
ap.add_argument("pmc")
ap.add_argument("stats")
ap.add_argument("-o", required=True)
a = ap.parse_args()
pmc = json.load(open(a.pmc))["kernels"]


def short(name):
    name = re.sub(r"^void\s+", "", name)
    depth, out = 0, []
    for ch in name:
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).strip()


dur = {}
for r in csv.DictReader(open(a.stats)):
    dur[short(r["Name"])] = (float(r["AverageNs"]), int(r["Calls"]), float(r["TotalDurationNs"]))
rows = {}
for k, v in pmc.items():
    if k not in dur or dur[k][0] <= 0:
        continue
    ns, calls, tot = dur[k]
    gbps = v["traffic_bytes_per_launch"] / ns          # bytes / ns = GB/s
    rows[k] = {"traffic_MB_per_launch": round(v["traffic_bytes_per_launch"] / 1e6, 2), "avg_us": round(ns / 1e3, 1),
               "calls_in_stats": calls, "total_ms_in_stats": round(tot / 1e6, 2), "hbm_GBps": round(gbps, 1),
               "frac_of_8TBps": round(gbps / 8000.0, 4)}
rows = dict(sorted(rows.items(), key=lambda kv: -kv[1]["total_ms_in_stats"]))
json.dump({"peak_GBps": 8000, "note": "traffic = PMC FETCH_SIZE (x2, gfx950) + WRITE_SIZE per launch, averaged over the "
           "launches of the PMC passes; duration = rocprofv3 --stats average of the stats pass (same workload, more steps)",
           "kernels": rows}, open(a.o, "w"), indent=1)
for k, v in list(rows.items())[:24]:
    print(f"{k[:60]:60s} {v['traffic_MB_per_launch']:9.1f} MB {v['avg_us']:8.1f} us {v['hbm_GBps']:8.0f} GB/s {v['frac_of_8TBps']:.3f}")
