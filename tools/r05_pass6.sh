#!/bin/bash
set -e -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
echo "[r05g] VAE tests"; date
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py tests/test_inference_gpu.py -x -q -k "vae or training_step_matches or recon_micro_batch" > "$OUT/r05g_tests.log" 2>&1 || { tail -40 "$OUT/r05g_tests.log"; exit 1; }
tail -3 "$OUT/r05g_tests.log"
echo "[r05g] VAE alone A/B"; date
for rep in 1 2; do for v in 1 0; do echo "downsample_bf16=$v $(ADAP_VAE_DOWNSAMPLE_BF16=$v timeout -k 10 300 python tools/vae_probe.py 2>&1 | tail -1)"; done; done
COMMON="--steps 20 --warmup 5 --no-cpu-baseline --no-distill-mix --no-ddim --no-unfrozen --no-compos --no-zs-frontend --no-roofline --no-rehearse-exchange --no-entry-leg"
for rep in 1 2; do
  for v in 1 0; do
    ADAP_VAE_DOWNSAMPLE_BF16=$v timeout -k 10 600 python bench.py $COMMON > "$OUT/r05g_bench_v${v}_$rep.log" 2>&1
    echo "downsample_bf16=$v rep=$rep $(tail -1 $OUT/r05g_bench_v${v}_$rep.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["host_work_ms_per_step"])')"
  done
done
echo "[r05g] smoke"; date
timeout -k 10 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "[r05g] done"; date
