#!/bin/bash
# MFMA-pipe utilisation of the attention and conv kernels from PMC counters:  bash tools/mfma_pmc.sh TAG  (via gpurun)
#   util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs)
# (SQ_VALU_MFMA_BUSY_CYCLES counts cycles, summed over the SIMDs; GRBM_GUI_ACTIVE is the sum over the 8 XCDs --
#  MI355X_MICROARCH.md, "DVFS give-back" and the PMC table).  Writes gpurun_out/<tag>_mfma_util.json.
set -e -o pipefail
TAG=${1:-mfma}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for probe in attn_probe conv_probe; do
  timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -f csv -d "$OUT/${TAG}_${probe}" -o "$TAG" -- python3 "$ROOT/tools/${probe}.py" > "$OUT/${TAG}_${probe}.log" 2>&1
done
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, sys, collections
out, tag = sys.argv[1], sys.argv[2]
res = {}
for probe in ("attn_probe", "conv_probe"):
    f = glob.glob(f"{out}/{tag}_{probe}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Grid_Size"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
    for k, v in acc.items():
        if not any(s in k[0] for s in ("attn_fwd", "attn_bwd", "conv3x3_halo", "conv_gemm_ring")):
            continue
        busy, gui = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), v.get("GRBM_GUI_ACTIVE", 0.0)
        if gui <= 0:
            continue
        util = busy / (gui / 8.0 * 256 * 4)
        res[f"{k[0]} grid={k[1]}"] = {"launches": n[(k, "GRBM_GUI_ACTIVE")], "mfma_busy_cycles": busy, "gui_active_sum_8xcd": gui,
                                      "mfma_pipe_util": round(util, 4)}
json.dump({"formula": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)", "kernels": res},
          open(f"{out}/{tag}_mfma_util.json", "w"), indent=1)
for k, v in res.items():
    print(f"{k[:80]:80s} util {v['mfma_pipe_util']:.3f}  ({v['launches']} launches)")
PY
