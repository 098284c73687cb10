#!/bin/bash
# two ranks of the bench's main step on ONE card over gloo (RCCL refuses two ranks on one device): the data-parallel code path with
# lanes -- reducer inside the lanes' gate, exchange after every micro-batch -- run end to end; a number of no meaning, a loss that must be finite
set -e -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
ADAP_DIST_BACKEND=gloo ADAP_GN_TWO_PASS=1 timeout -k 10 900 python bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline --no-distill-mix --no-ddim --no-unfrozen --no-compos --no-zs-frontend --no-roofline > "$OUT/r05_two_rank_rehearsal.log" 2>&1
tail -1 "$OUT/r05_two_rank_rehearsal.log" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ("value","n_gpus","ms_per_step","micro_batch_lanes","final_loss")}, d["config"]["dist_backend"], d["config"]["grad_allreduce_bytes"])'
