#!/bin/bash
# DESIGN 3f's model, from a kernel trace: every launch of the main step (each kernel alone on the chip: --no-lanes --no-prefetch) is
# classed by its grid -- "fills the chip" = at least one workgroup per CU (>= 256) -- and the kernel time of both classes is summed per
# micro-batch.  The model: step ~ fill + rest / 2 (two lanes overlap only what leaves CUs idle).
set -e -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-distill-mix --no-ddim --no-unfrozen --no-compos --no-zs-frontend --no-rehearse-exchange --no-entry-leg --no-roofline"
timeout -k 10 500 rocprofv3 --kernel-trace -f csv -d "$OUT/r05_trace0" -o r05 -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 --no-lanes --no-prefetch $COMMON > "$OUT/r05_trace0.log" 2>&1
cd "$ROOT"
python3 - <<'PY'
import csv, glob, json, sys, collections
csv.field_size_limit(sys.maxsize)
f = glob.glob("gpurun_out/r05_trace0/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
STEPS = 16                               # 3 warm-up + 10 timed + 3 host-probe micro-batches (no roofline pass)
cls = collections.defaultdict(lambda: [0, 0.0])
fam = collections.defaultdict(lambda: [0, 0.0, 0, 0.0])
for r in rows:
    wg = max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
    nwg = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // wg
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    name = r["Kernel_Name"]
    third = any(s in name for s in ("Cijk", "at::", "rocclr"))
    key = "third-party" if third else ("fills" if nwg >= 256 else "leaves CUs idle")
    cls[key][0] += 1
    cls[key][1] += us
    short = name.split("(")[0].replace("void ", "")[:48]
    e = fam[short]
    if nwg >= 256:
        e[0] += 1; e[1] += us
    else:
        e[2] += 1; e[3] += us
res = {k: {"launches_per_micro_batch": round(n / STEPS, 1), "kernel_ms_per_micro_batch": round(t / 1e3 / STEPS, 3)} for k, (n, t) in cls.items()}
fill, idle, third = (res.get(k, {"kernel_ms_per_micro_batch": 0})["kernel_ms_per_micro_batch"] for k in ("fills", "leaves CUs idle", "third-party"))
res["model_step_ms"] = {"fill + idle / 2 + third-party": round(fill + idle / 2 + third, 2), "fill + idle / 2": round(fill + idle / 2, 2)}
top = sorted(fam.items(), key=lambda kv: -(kv[1][1] + kv[1][3]))[:24]
res["by_kernel"] = {k: {"fills": [round(v[0] / STEPS, 1), round(v[1] / 1e3 / STEPS, 3)], "idle": [round(v[2] / STEPS, 1), round(v[3] / 1e3 / STEPS, 3)]} for k, v in top}
json.dump({"what": "kernel time per micro-batch of the main step, each kernel alone on the chip (--no-lanes --no-prefetch), classed by grid: "
                   "'fills' = >= 256 workgroups (one per CU).  by_kernel: [launches, ms] per micro-batch in either class (includes the "
                   "process's one-time launches / 16)", **res}, open("gpurun_out/r05_step_model.json", "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "by_kernel"}, indent=1))
PY
rm -rf "$OUT/r05_trace0"
