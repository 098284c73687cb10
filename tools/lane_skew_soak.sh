#!/bin/bash
# One process, the lanes skewed on purpose: ADAP_DIAG_LANE_DELAY="lane,us" parks an idle one-workgroup kernel on that lane at the start of
# each of its contexts (forward and backward of every window), so one lane runs tens of milliseconds behind the other.  The losses must
# stay what they are without the delay (to the stand-in's stream-K jitter, ~1e-4): a dependency the lanes' events do not express would
# change them.  (It does NOT reproduce the first-window pack race -- that one needed a second process, tools/lanes_two_process_soak.sh.)   bash tools/lane_skew_soak.sh
C="--gpus 1 --steps 8 --warmup 2 --no-cpu-baseline --no-distill-mix --no-ddim --no-unfrozen --no-compos --no-zs-frontend --no-roofline --no-rehearse-exchange --no-entry-leg $1"
P='import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["final_loss"])'
for cfg in "X=1" "ADAP_DIAG_LANE_DELAY=0,20000" "ADAP_DIAG_LANE_DELAY=1,20000" "ADAP_DIAG_LANE_DELAY=0,3000" "ADAP_DIAG_LANE_DELAY=1,3000" "ADAP_DIAG_LANE_DELAY=1,60000"; do
  for i in 1 2; do echo -n "$cfg: "; env $cfg ADAP_GN_TWO_PASS=1 timeout -k 10 300 python bench.py $C 2>/dev/null | tail -1 | python -c "$P"; done
done
