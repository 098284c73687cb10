#!/bin/bash
set -e -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-distill-mix --no-ddim --no-unfrozen --no-compos --no-zs-frontend --no-rehearse-exchange --no-entry-leg"
echo "[m2] stats pass, lanes"; date
timeout -k 10 500 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/r05_stats_l" -o r05 -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 --no-clock-probe --no-aggregates $COMMON > "$OUT/r05_stats_l.log" 2>&1
find "$OUT/r05_stats_l" -name "*kernel_trace.csv" -delete || true
cd "$ROOT"
echo "[m2] mfma"; date
bash tools/mfma_pmc.sh r05 2>&1 | tail -20
echo "[m2] extras"; date
bash tools/measure_r05_extra.sh
