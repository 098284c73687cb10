#!/bin/bash
# round 5, pass 3: batched token maps (parity + A/B), attention bwd XCD rule, Lightning entry with two windows ahead
set -e -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
echo "[r05c] tests"; date
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py tests/test_regloss_gpu.py -x -q -k "token_map or tokmap or hoist or transformer_block or unet_narrow or recon_step or training_step_matches or attn" > "$OUT/r05c_tests.log" 2>&1 || { tail -40 "$OUT/r05c_tests.log"; exit 1; }
tail -3 "$OUT/r05c_tests.log"
echo "[r05c] bench A/B"; date
COMMON="--steps 20 --warmup 5 --no-cpu-baseline --no-distill-mix --no-ddim --no-unfrozen --no-compos --no-zs-frontend --no-roofline --no-rehearse-exchange --no-entry-leg"
for rep in 1 2; do
  for h in 1 0; do
    ADAP_BATCH_TOKMAPS=$h timeout -k 10 600 python bench.py $COMMON > "$OUT/r05c_bench_t${h}_$rep.log" 2>&1
    echo "batch_tokmaps=$h rep=$rep $(tail -1 $OUT/r05c_bench_t${h}_$rep.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["host_work_ms_per_step"])')"
  done
done
echo "[r05c] full bench legs"; date
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-distill-mix --no-ddim --no-unfrozen --no-compos --no-zs-frontend > "$OUT/r05c_bench_full.log" 2>&1
tail -1 "$OUT/r05c_bench_full.log" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"]); print(json.dumps(d.get("exchange_rehearsal"))); print(json.dumps(d.get("entry_lightning")))'
echo "[r05c] done"; date
