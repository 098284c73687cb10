#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (counter_collection.csv) into per-kernel HBM traffic.

    python tools/pmc_summary.py --fetch DIR_WITH_FETCH_SIZE_PASS --write DIR_WITH_WRITE_SIZE_PASS \
        [--calib sumsq_partial_kernel:BYTES] -o profiles/rNN_pmc_traffic.json

FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950 (TCC slots), so they come from two runs of the same
command.  Units / corrections as /opt/skills/guides/MI355X_MICROARCH.md "HBM" prescribes: both counters are in KB;
on gfx950 FETCH_SIZE reports HALF the bytes of 16-B-per-lane streaming reads (global_load and buffer_load...lds
alike) -> doubled; WRITE_SIZE is exact for 16-B-per-lane stores.  ``--calib name:bytes`` checks the correction on a
kernel of this very run whose read volume is known (e.g. the optimiser's sum-of-squares pass over the flat gradient
buffer) and records the ratio.  Kernel names are shortened to the symbol up to its argument list."""
import argparse
import csv
import glob
import json
import os
import re
import sys

csv.field_size_limit(sys.maxsize)


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


def load(d, counter):
    per = {}
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                k = short(row["Kernel_Name"])
                e = per.setdefault(k, {"launches": 0, "kb": 0.0, "by_grid": {}})
                v = float(row["Counter_Value"])
                e["launches"] += 1
                e["kb"] += v
                g = e["by_grid"].setdefault(row["Grid_Size"], [0, 0.0])
                g[0] += 1
                g[1] += v
    return per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--calib", default="")
    ap.add_argument("--command", default="")
    ap.add_argument("--only", default="conv,attn,gn_,ln_,splitk,prodigy,sumsq,geglu")
    ap.add_argument("-o", "--out", required=True)
    a = ap.parse_args()
    F, W = load(a.fetch, "FETCH_SIZE"), load(a.write, "WRITE_SIZE")
    keep = [s for s in a.only.split(",") if s]
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two passes, --kernel-trace only)",
           "command": a.command,
           "correction": "bytes_read = 2 * FETCH_SIZE[KB] * 1024 (gfx950, 16 B/lane reads); bytes_written = WRITE_SIZE[KB] * 1024",
           "kernels": {}}
    for k in sorted(F):
        if keep and not any(s in k for s in keep):
            continue
        f, w = F[k], W.get(k, {"launches": 0, "kb": 0.0, "by_grid": {}})
        rd = 2.0 * 1024.0 * f["kb"] / f["launches"]
        wr = 1024.0 * w["kb"] / w["launches"] if w["launches"] else 0.0
        grids = {}
        for g, (n, kb) in f["by_grid"].items():
            wn, wkb = w["by_grid"].get(g, [0, 0.0])
            grids[g] = {"launches": n, "read_MB": round(2.0 * 1024 * kb / n / 1e6, 2),
                        "written_MB": round(1024 * wkb / wn / 1e6, 2) if wn else None}
        out["kernels"][k] = {"launches": f["launches"], "read_bytes_per_launch": round(rd),
                             "written_bytes_per_launch": round(wr), "traffic_bytes_per_launch": round(rd + wr),
                             "by_grid_size": grids}
    if a.calib:
        name, known = a.calib.rsplit(":", 1)
        hit = [k for k in F if name in k]
        if hit:
            k = hit[0]
            rd = 2.0 * 1024.0 * F[k]["kb"] / F[k]["launches"]
            out["calibration"] = {"kernel": k, "known_read_bytes": int(known), "corrected_read_bytes": round(rd),
                                  "ratio": round(rd / float(known), 4)}
    with open(a.out, "w") as fh:
        json.dump(out, fh, indent=1)
    print(f"wrote {a.out}: {len(out['kernels'])} kernels", out.get("calibration", ""))


if __name__ == "__main__":
    main()
