#!/bin/bash
# A/B of environment switches on the default bench loop: bash tools/ab_env.sh "VAR=1" "OTHER=2" ... (each run twice, interleaved)
B="python bench.py --no-cpu-baseline --no-distill-mix --no-ddim --no-unfrozen --no-compos --no-zs-frontend --no-roofline --no-rehearse-exchange --no-entry-leg --steps 20 --warmup 4"
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["host_work_ms_per_step"])'
for rep in 1 2; do
  for cfg in "BASE=1" "$@"; do
    echo -n "$cfg  "; env $cfg ADAP_BENCH_WATCHDOG=200 timeout -k 10 300 $B 2>>gpurun_out/ab_env_stderr.log | tail -1 | python -c "$P"
  done
done
