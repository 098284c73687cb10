#!/bin/bash
# round 5, first GPU pass: the new window / data-parallel tests, attention parity, and the attention A/B (XCD map off / on)
set -e -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
echo "[r05] window + parallel tests"; date
timeout -k 10 900 python -m pytest tests/test_windows_gpu.py tests/test_parallel_gpu.py -x -q > "$OUT/r05a_tests_windows.log" 2>&1 || { tail -40 "$OUT/r05a_tests_windows.log"; exit 1; }
tail -3 "$OUT/r05a_tests_windows.log"
echo "[r05] attention + lanes tests"; date
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "attn or attention" > "$OUT/r05a_tests_attn.log" 2>&1 || { tail -40 "$OUT/r05a_tests_attn.log"; exit 1; }
tail -3 "$OUT/r05a_tests_attn.log"
timeout -k 10 900 python -m pytest tests/test_model_gpu.py -x -q -k "window or lanes or training" > "$OUT/r05a_tests_model.log" 2>&1 || { tail -40 "$OUT/r05a_tests_model.log"; exit 1; }
tail -3 "$OUT/r05a_tests_model.log"
echo "[r05] attention A/B timings"; date
ADAP_ATTN_XCD=0 timeout -k 10 300 python tools/attn_xcd_probe.py > "$OUT/r05a_attn_xcd0.log" 2>&1
ADAP_ATTN_XCD=1 timeout -k 10 300 python tools/attn_xcd_probe.py > "$OUT/r05a_attn_xcd1.log" 2>&1
cat "$OUT/r05a_attn_xcd0.log" "$OUT/r05a_attn_xcd1.log"
echo "[r05] attention FETCH_SIZE passes"; date
cd /tmp && export TMPDIR=/tmp
export ITERS=5
ADAP_ATTN_XCD=0 timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d "$OUT/r05a_pmcF_xcd0" -o x -- python3 "$ROOT/tools/attn_xcd_probe.py" > "$OUT/r05a_pmcF_xcd0.log" 2>&1
ADAP_ATTN_XCD=1 timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d "$OUT/r05a_pmcF_xcd1" -o x -- python3 "$ROOT/tools/attn_xcd_probe.py" > "$OUT/r05a_pmcF_xcd1.log" 2>&1
find "$OUT/r05a_pmcF_xcd0" "$OUT/r05a_pmcF_xcd1" -name "*kernel_trace.csv" -delete || true
cd "$ROOT"
echo "[r05] bench"; date
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-distill-mix --no-ddim --no-unfrozen --no-compos --no-zs-frontend > "$OUT/r05a_bench.log" 2>&1
tail -1 "$OUT/r05a_bench.log" | cut -c1-600
echo "[r05] done"; date
