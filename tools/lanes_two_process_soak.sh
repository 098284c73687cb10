#!/bin/bash
# Soak of the two-lane loop with TWO processes on one card (they share the CUs, so the lanes of each drift apart far more than alone):
# the final loss of every run must be finite and repeat to ~1e-4 (the stand-in's hipBLASLt stream-K GEMM is the only source of jitter).
# This is how the first-window race on lazily built weight packs was found (ops.note_cache_fill; 25 % of runs ended in NaN).
#   bash tools/lanes_two_process_soak.sh "<extra bench args>" "<ENV=1 ...>" N
# ADAP_DIAG_NO_EXCHANGE=1 in the second argument: the ranks do not exchange gradients (50 ms per step instead of seconds over gloo);
# ADAP_DIAG_NO_FILL_DRAIN=1 switches the fix off again (NaN in 1 of 16 runs on the day it was written; 0 of 80 with it);
# ADAP_DIAG_WARM_ONE_STREAM=1 runs one window on one stream first (every cache filled before the lanes start): with it and WITHOUT the
# drain 80 of 80 runs were finite -- the race is confined to first-use fills, the steady state is clean.
C="--gpus 2 --steps 6 --warmup 2 --no-cpu-baseline --no-distill-mix --no-ddim --no-unfrozen --no-compos --no-zs-frontend --no-roofline $1"
for i in $(seq 1 ${3:-8}); do
  env ${2:-X=1} ADAP_DIST_BACKEND=gloo ADAP_GN_TWO_PASS=1 timeout -k 10 300 python bench.py $C 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["final_loss"])'
done
