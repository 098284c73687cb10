#!/bin/bash
set -e -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
echo "[r05j] kernel tests"; date
timeout -k 10 1000 python -m pytest tests/test_kernels_gpu.py tests/test_wgrad_gpu.py -x -q > "$OUT/r05j_tests.log" 2>&1 || { tail -40 "$OUT/r05j_tests.log"; exit 1; }
tail -3 "$OUT/r05j_tests.log"
echo "[r05j] ff probe, wide stores on / off"; date
ITERS=100 timeout -k 10 300 python tools/ff_probe.py 2>&1 | grep -E "^---|auto"
ADAP_CONV_DEBUG=8 ITERS=100 timeout -k 10 300 python tools/ff_probe.py 2>&1 | grep -E "^---|auto"
echo "[r05j] bench A/B"; date
COMMON="--steps 20 --warmup 5 --no-cpu-baseline --no-distill-mix --no-ddim --no-unfrozen --no-compos --no-zs-frontend --no-roofline --no-rehearse-exchange --no-entry-leg"
for rep in 1 2; do
  for v in 0 8; do
    ADAP_CONV_DEBUG=$v timeout -k 10 600 python bench.py $COMMON > "$OUT/r05j_bench_d${v}_$rep.log" 2>&1
    echo "conv_debug=$v (8 = narrow stores) rep=$rep $(tail -1 $OUT/r05j_bench_d${v}_$rep.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["final_loss"])')"
  done
done
echo "[r05j] done"; date
