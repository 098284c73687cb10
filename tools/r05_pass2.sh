#!/bin/bash
# round 5, pass 2: context K|V hoist parity + A/B, attention XCD map per kernel
set -e -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
echo "[r05b] hoist + block tests"; date
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py -x -q -k "hoist or transformer_block or unet_narrow or sd15_full" > "$OUT/r05b_tests.log" 2>&1 || { tail -40 "$OUT/r05b_tests.log"; exit 1; }
tail -3 "$OUT/r05b_tests.log"
echo "[r05b] attention XCD bitmask"; date
for m in 0 1 3 5 7; do
  ADAP_ATTN_XCD=$m ITERS=200 timeout -k 10 300 python tools/attn_xcd_probe.py 2>/dev/null | grep -E "ADAP|64x64|32x32 " 
done | tee "$OUT/r05b_attn_xcd_bits.log"
echo "[r05b] bench A/B"; date
COMMON="--steps 20 --warmup 5 --no-cpu-baseline --no-distill-mix --no-ddim --no-unfrozen --no-compos --no-zs-frontend --no-roofline --no-rehearse-exchange --no-entry-leg"
for rep in 1 2; do
  for h in 1 0; do
    ADAP_HOIST_KV=$h timeout -k 10 600 python bench.py $COMMON > "$OUT/r05b_bench_h${h}_$rep.log" 2>&1
    echo "hoist=$h rep=$rep $(tail -1 $OUT/r05b_bench_h${h}_$rep.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["host_work_ms_per_step"])')"
  done
done
echo "[r05b] full bench legs"; date
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-distill-mix --no-ddim --no-unfrozen --no-compos --no-zs-frontend > "$OUT/r05b_bench_full.log" 2>&1
tail -1 "$OUT/r05b_bench_full.log" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"]); print(json.dumps(d.get("exchange_rehearsal"))); print(json.dumps(d.get("entry_lightning")))'
echo "[r05b] done"; date
