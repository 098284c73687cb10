#!/bin/bash
# rocprofv3 --stats of an arbitrary tools/*.py script:  bash tools/prof_script.sh TAG tools/vae_probe.py [args]
set -e -o pipefail
TAG=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
SCRIPT=$ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/${TAG}_stats" -o "$TAG" -- python3 "$SCRIPT" "$@" > "$OUT/${TAG}.log" 2>&1
find "$OUT/${TAG}_stats" -name "*kernel_trace.csv" -delete || true
tail -2 "$OUT/${TAG}.log"
