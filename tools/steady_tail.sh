#!/bin/bash
# The third-party tail of ONE steady-state micro-batch: a kernel trace of the main step with every kernel alone
# on the chip (--no-lanes --no-prefetch), cut at the VAE's first convolution (conv3x3_rgb_kernel: exactly one launch per
# micro-batch) so that the process's one-time launches (random weights, the first packs) do not count.
#   bash tools/steady_tail.sh [tag]     -> gpurun_out/<tag>_steady_tail.json
set -e -o pipefail
TAG=${1:-r05}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-distill-mix --no-ddim --no-unfrozen --no-compos --no-zs-frontend --no-rehearse-exchange --no-entry-leg --no-roofline --no-clock-probe --no-aggregates"
timeout -k 10 500 rocprofv3 --kernel-trace -f csv -d "$OUT/${TAG}_tail_trace" -o "$TAG" -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 --no-lanes --no-prefetch $COMMON > "$OUT/${TAG}_tail_trace.log" 2>&1
cd "$ROOT"
python3 tools/steady_tail.py "$OUT/${TAG}_tail_trace" "$OUT/${TAG}_steady_tail.json"
rm -rf "$OUT/${TAG}_tail_trace"
