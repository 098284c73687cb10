#!/bin/bash
set -e -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
COMMON="--steps 20 --warmup 5 --no-cpu-baseline --no-distill-mix --no-ddim --no-unfrozen --no-compos --no-zs-frontend --no-roofline --no-rehearse-exchange --no-entry-leg"
echo "[r05e] lane skew A/B"; date
for rep in 1 2; do
  for L in -1 2 5 8 12 16; do
    ADAP_LANES_SKEW_LAYER=$L timeout -k 10 600 python bench.py $COMMON > "$OUT/r05e_bench_s${L}_$rep.log" 2>&1
    echo "skew_layer=$L rep=$rep $(tail -1 $OUT/r05e_bench_s${L}_$rep.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["host_work_ms_per_step"], d["final_loss"])')"
  done
done
echo "[r05e] done"; date
