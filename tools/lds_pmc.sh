#!/bin/bash
# LDS bank-conflict counters of the conv probe shapes:  bash tools/lds_pmc.sh TAG   (run through gpurun)
set -e -o pipefail
TAG=${1:-lds}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -f csv -d "$OUT/${TAG}_lds" -o "$TAG" -- python3 "$ROOT/tools/conv_probe.py" > "$OUT/${TAG}_lds.log" 2>&1
python3 - "$OUT/${TAG}_lds" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = (r["Kernel_Name"][:60], r["Grid_Size"])
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[k] += 1
for k, v in acc.items():
    if "conv" in k[0]:
        a, c = v.get("SQ_LDS_IDX_ACTIVE", 0), v.get("SQ_LDS_BANK_CONFLICT", 0)
        print(k, "active", a, "conflict", c, "ratio %.3f" % (c / a if a else 0))
PY
