#!/bin/bash
# What the attention kernels' SIMDs spend their cycles on, from PMC counters:  bash tools/valu_pmc.sh TAG  (via gpurun)
# Three counter passes over tools/attn_probe.py (separate runs; --pmc with --kernel-trace only):
#   A  SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE          matrix-pipe busy share  = busy / (gui / 8 XCDs * 1024 SIMDs)
#   B  SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE
#                                                        vector issue share (SQ_ACTIVE_INST_VALU is in quad-cycles, summed over waves)
#                                                        and vector instructions per MFMA
#   C  SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_MFMA   transcendental / convert instructions per MFMA
# Writes gpurun_out/<tag>_valu_pmc.json.
set -e -o pipefail
TAG=${1:-valu}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -f csv -d "$OUT/${TAG}_vA" -o "$TAG" -- python3 "$ROOT/tools/attn_probe.py" > "$OUT/${TAG}_vA.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace -f csv -d "$OUT/${TAG}_vB" -o "$TAG" -- python3 "$ROOT/tools/attn_probe.py" > "$OUT/${TAG}_vB.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_MFMA --kernel-trace -f csv -d "$OUT/${TAG}_vC" -o "$TAG" -- python3 "$ROOT/tools/attn_probe.py" > "$OUT/${TAG}_vC.log" 2>&1
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, sys, collections
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for p in "ABC":
    f = glob.glob(f"{out}/{tag}_v{p}/**/*counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Grid_Size"])
        if not any(s in k[0] for s in ("attn_fwd", "attn_bwd_dq", "attn_bwd_dkv")):
            continue
        acc[k][p + ":" + r["Counter_Name"]] += float(r["Counter_Value"])
res = {}
for k, v in acc.items():
    simd_cycles_a = v.get("A:GRBM_GUI_ACTIVE", 0) / 8.0 * 1024
    simd_cycles_b = v.get("B:GRBM_GUI_ACTIVE", 0) / 8.0 * 1024
    if simd_cycles_a <= 0 or simd_cycles_b <= 0 or v.get("B:SQ_INSTS_MFMA", 0) <= 0:
        continue
    res[f"{k[0]} grid={k[1]}"] = {
        "mfma_pipe_busy_share": round(v["A:SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles_a, 4),
        "valu_issue_share": round(4.0 * v["B:SQ_ACTIVE_INST_VALU"] / simd_cycles_b, 4),
        "valu_insts_per_mfma": round(v["B:SQ_INSTS_VALU"] / v["B:SQ_INSTS_MFMA"], 2),
        "transcendental_per_mfma": round(v.get("C:SQ_INSTS_VALU_TRANS_F32", 0) / max(v.get("C:SQ_INSTS_MFMA", 0), 1), 2),
        "convert_per_mfma": round(v.get("C:SQ_INSTS_VALU_CVT", 0) / max(v.get("C:SQ_INSTS_MFMA", 0), 1), 2)}
json.dump({"formulas": {"mfma_pipe_busy_share": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)",
                        "valu_issue_share": "4 * SQ_ACTIVE_INST_VALU (quad-cycles over all waves) / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs); includes the MFMAs' own issue",
                        "valu_insts_per_mfma": "SQ_INSTS_VALU / SQ_INSTS_MFMA (SQ_INSTS_VALU counts the MFMAs too)"},
           "kernels": res}, open(f"{out}/{tag}_valu_pmc.json", "w"), indent=1)
for k, v in res.items():
    print(f"{k[:60]:60s} " + "  ".join(f"{a} {b}" for a, b in v.items()))
PY
