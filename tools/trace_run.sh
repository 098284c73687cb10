#!/bin/bash
# rocprofv3 kernel TRACE (start/end per launch) of a short bench run -> gpurun_out/<tag>_trace/<tag>_kernel_trace.csv
set -e -o pipefail
TAG=${1:-tr}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace -f csv -d "$OUT/${TAG}_trace" -o "$TAG" -- python3 "$ROOT/bench.py" --steps 6 --warmup 3 --no-cpu-baseline --no-distill-mix --no-ddim --no-roofline --no-unfrozen --no-compos --no-zs-frontend --no-clock-probe > "$OUT/${TAG}_trace.log" 2>&1
ls -la "$OUT/${TAG}_trace"/*/ || ls -la "$OUT/${TAG}_trace"
grep '"metric"' "$OUT/${TAG}_trace.log" | cut -c1-200
