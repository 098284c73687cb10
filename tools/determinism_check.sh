#!/bin/bash
# the same bench command several times: the final loss of a run is a function of the seeds alone (no atomics, fixed-order reductions),
# so it must repeat digit for digit -- one rank, one rank with two-pass GroupNorms, two ranks on one card over gloo
C="--steps 6 --warmup 2 --no-cpu-baseline --no-distill-mix --no-ddim --no-unfrozen --no-compos --no-zs-frontend --no-roofline --no-rehearse-exchange --no-entry-leg"
P='import json,sys; d=json.loads(sys.stdin.read()); print(d["n_gpus"], d["ms_per_step"], d["final_loss"])'
for i in 1 2 3; do echo -n "one rank: "; timeout -k 10 300 python bench.py --gpus 1 $C 2>/dev/null | tail -1 | python -c "$P"; done
for i in 1 2 3; do echo -n "one rank, two-pass GN: "; ADAP_GN_TWO_PASS=1 timeout -k 10 300 python bench.py --gpus 1 $C 2>/dev/null | tail -1 | python -c "$P"; done
for i in 1 2 3; do echo -n "two ranks (gloo, one card): "; ADAP_DIST_BACKEND=gloo ADAP_GN_TWO_PASS=1 timeout -k 10 300 python bench.py --gpus 2 $C 2>/dev/null | tail -1 | python -c "$P"; done
