#!/bin/bash
# Where the N 4096 attention kernels' cycles go (PMC, separate passes; tools/attn_quick_probe.py under the counters):
#   bash tools/attn_pmc.sh TAG   -> gpurun_out/<TAG>_attn_pmc.json  (per kernel, per launch: counter sums over the chip)
set -e -o pipefail
TAG=${1:-attnpmc}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_LDS_ADDR_CONFLICT"; do
  i=$((i + 1))
  ADAP_PROBE_ITERS=10 timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace -f csv -d "$OUT/${TAG}_p$i" -o "$TAG" -- python3 "$ROOT/tools/attn_quick_probe.py" > "$OUT/${TAG}_p$i.log" 2>&1
done
cd "$ROOT"
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, sys, collections
csv.field_size_limit(sys.maxsize)
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for f in glob.glob(f"{out}/{tag}_p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "attn_" not in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
res = {k: {c: round(v / n[(k, c)], 1) for c, v in cs.items()} for k, cs in acc.items()}
json.dump(res, open(f"{out}/{tag}_attn_pmc.json", "w"), indent=1)
for k, cs in res.items():
    print(k)
    wc = cs.get("SQ_WAVE_CYCLES", 0) or 1
    for c, v in sorted(cs.items()):
        print(f"    {c:28s} {v:16.0f}  {v / wc:7.3f} of SQ_WAVE_CYCLES")
PY
find "$OUT" -path "*${TAG}_p*" -name "*kernel_trace.csv" -delete || true
