#!/bin/bash
# rocprofv3 kernel stats of a short bench run -> gpurun_out/<tag>_stats/<tag>_kernel_stats.csv ; prints the bench line
set -e -o pipefail
TAG=${1:-t}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/${TAG}_stats" -o "$TAG" -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-distill-mix --no-ddim --no-roofline --no-unfrozen --no-compos --no-zs-frontend > "$OUT/${TAG}_stats.log" 2>&1
find "$OUT/${TAG}_stats" -name "*kernel_trace.csv" -delete || true
grep '"metric"' "$OUT/${TAG}_stats.log" | cut -c1-220
