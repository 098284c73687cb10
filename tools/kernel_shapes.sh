#!/bin/bash
# Which launches hide behind a kernel name: a kernel trace of the main step (alone: --no-lanes --no-prefetch), steady-state micro-batches
# only, grouped by (kernel, grid, workgroup): launches per micro-batch, average and total duration.   bash tools/kernel_shapes.sh PATTERN
set -e -o pipefail
PAT=${1:-conv_gemm_kernel}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-distill-mix --no-ddim --no-unfrozen --no-compos --no-zs-frontend --no-rehearse-exchange --no-entry-leg --no-roofline --no-clock-probe --no-aggregates"
timeout -k 10 500 rocprofv3 --kernel-trace -f csv -d "$OUT/shapes_trace" -o shapes -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 --no-lanes --no-prefetch $COMMON > "$OUT/shapes_trace.log" 2>&1
cd "$ROOT"
python3 - "$OUT/shapes_trace" "$PAT" <<'PY'
import csv, glob, sys, collections
csv.field_size_limit(sys.maxsize)
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "conv3x3_rgb_kernel" in r["Kernel_Name"]]
seg = rows[marks[5]:marks[11]]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in seg:
    if sys.argv[2] in r["Kernel_Name"]:
        wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
        g = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]) // int(r["Workgroup_Size_Y"]), int(r["Grid_Size_Z"]) // int(r["Workgroup_Size_Z"]))
        k = (r["Kernel_Name"].split("(")[0].replace("void ", ""), g, r.get("LDS_Block_Size", ""))
        acc[k][0] += 1
        acc[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{n / 6:6.1f} x {t / n:7.1f} us = {t / 6:8.1f} us/mb  {k}")
PY
rm -rf "$OUT/shapes_trace"
