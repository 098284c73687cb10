"""Steady-state census of a kernel trace (tools/steady_tail.sh): the launches between the VAE's first convolutions of mid-run
micro-batches, split into the package's own kernels and the third-party ones (aten / rocclr / hipBLASLt / rocprim)."""
import collections
import csv
import glob
import json
import re
import sys

csv.field_size_limit(sys.maxsize)
src, dst = sys.argv[1], sys.argv[2]
f = glob.glob(src + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "conv3x3_rgb_kernel" in r["Kernel_Name"]]
if len(marks) < 13:
    sys.exit(f"only {len(marks)} micro-batch markers in the trace")
FIRST, N = 5, 6                                  # micro-batches 5 .. 10 of 16: inside the timed region, three optimiser steps
seg = rows[marks[FIRST]:marks[FIRST + N]]
THIRD = re.compile(r"Cijk|at::|rocclr|rocprim|hipcub|thrust|elementwise_kernel_with_index")
fam = collections.defaultdict(lambda: [0, 0.0])
tot = {"own": [0, 0.0], "third_party": [0, 0.0], "stand_in_gemm": [0, 0.0]}
for r in seg:
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    name = r["Kernel_Name"]
    if THIRD.search(name):
        cls = "stand_in_gemm" if "Cijk" in name else "third_party"
        short = re.sub(r"\s+", " ", name.replace("void ", ""))[:110]
        fam[short][0] += 1
        fam[short][1] += us
    else:
        cls = "own"
    tot[cls][0] += 1
    tot[cls][1] += us
res = {"what": f"{N} steady-state micro-batches (markers {FIRST}..{FIRST + N} of {len(marks)}) of bench.py --no-lanes --no-prefetch; "
               "per micro-batch: [launches, kernel ms]",
       "per_micro_batch": {k: [round(n / N, 1), round(t / 1e3 / N, 3)] for k, (n, t) in tot.items()},
       "third_party_by_kernel": {k: [round(n / N, 2), round(t / N, 1)] for k, (n, t) in sorted(fam.items(), key=lambda kv: -kv[1][1])}}
json.dump(res, open(dst, "w"), indent=1)
print(json.dumps(res["per_micro_batch"]))
for k, v in list(res["third_party_by_kernel"].items())[:40]:
    print(f"{v[0]:7.2f} {v[1]:8.1f} us  {k}")
