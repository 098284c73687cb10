#!/bin/bash
# One measurement pass on the GPU box (run through gpurun from the repo root):
#   bash tools/measure_round.sh r01
# -> gpurun_out/<tag>_stats/  rocprofv3 --kernel-trace --stats of the default bench command (two lanes + prefetch stream)
#    gpurun_out/<tag>_stats1/ the same with --no-lanes (one stream + prefetch), <tag>_stats0/ with --no-lanes --no-prefetch (alone)
#    gpurun_out/<tag>_pmcF|W/ separate --pmc FETCH_SIZE / WRITE_SIZE passes (3 timed steps)
#    gpurun_out/<tag>_bench_line.json  the plain bench line
# Copy what is to be judged into profiles/ afterwards; the traffic summary:
#   python tools/pmc_summary.py --fetch gpurun_out/<tag>_pmcF --write gpurun_out/<tag>_pmcW --calib sumsq_partial_kernel:596353024 \
#          -o profiles/<tag>_pmc_traffic.json
set -e -o pipefail
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-distill-mix --no-ddim --no-unfrozen --no-compos --no-zs-frontend --no-rehearse-exchange --no-entry-leg"   # the profiled legs: the main step only
echo "[measure] stats pass"; date
timeout -k 10 500 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/${TAG}_stats" -o "$TAG" -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 --no-clock-probe --no-aggregates $COMMON > "$OUT/${TAG}_stats.log" 2>&1
echo "[measure] stats pass, one stream (--no-lanes: the kernels without the other lane's contention, as bench.py's HIP-event roofline times them)"; date
timeout -k 10 500 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/${TAG}_stats1" -o "$TAG" -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 --no-lanes --no-clock-probe --no-aggregates $COMMON > "$OUT/${TAG}_stats1.log" 2>&1
echo "[measure] stats pass, every kernel alone on the chip (--no-lanes --no-prefetch)"; date
timeout -k 10 500 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/${TAG}_stats0" -o "$TAG" -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 --no-lanes --no-prefetch --no-clock-probe --no-aggregates $COMMON > "$OUT/${TAG}_stats0.log" 2>&1
echo "[measure] pmc FETCH_SIZE pass"; date
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d "$OUT/${TAG}_pmcF" -o "$TAG" -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-roofline $COMMON > "$OUT/${TAG}_pmcF.log" 2>&1
echo "[measure] pmc WRITE_SIZE pass"; date
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d "$OUT/${TAG}_pmcW" -o "$TAG" -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-roofline $COMMON > "$OUT/${TAG}_pmcW.log" 2>&1
echo "[measure] plain bench"; date
cd "$ROOT"
timeout -k 10 600 python3 bench.py > "$OUT/${TAG}_bench.log" 2>&1
tail -1 "$OUT/${TAG}_bench.log" > "$OUT/${TAG}_bench_line.json"
echo "[measure] plain bench at the driver's arguments of earlier rounds (--steps 20 --warmup 5), main step only"; date
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > "$OUT/${TAG}_bench_driver_args.log" 2>&1
tail -1 "$OUT/${TAG}_bench_driver_args.log" > "$OUT/${TAG}_bench_line_driver_args.json"
# keep the merge small: the raw traces are large, the stats / counter csv are what is summarised
find "$OUT/${TAG}_stats" "$OUT/${TAG}_stats1" "$OUT/${TAG}_stats0" -name "*kernel_trace.csv" -delete || true
find "$OUT/${TAG}_pmcF" "$OUT/${TAG}_pmcW" -name "*kernel_trace.csv" -delete || true
du -sh "$OUT/${TAG}_stats" "$OUT/${TAG}_pmcF" "$OUT/${TAG}_pmcW"
echo "[measure] done"; date
