#!/bin/bash
# config 2's distillation mix leg under env switches / bench flags: bash tools/ab_distill.sh "VAR=1 --flag" ...
B="python bench.py --no-cpu-baseline --no-ddim --no-unfrozen --no-compos --no-zs-frontend --no-roofline --no-aggregates --steps 6 --warmup 2"
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["config2_distill_mix"]["images_per_sec"], d["config2_distill_mix"]["ms_per_micro_batch"])'
for rep in 1 2; do
  for cfg in "X=1" "$@"; do
    envs=""; flags=""
    for w in $cfg; do case "$w" in --*) flags="$flags $w";; *) envs="$envs $w";; esac; done
    echo -n "[$cfg]  "; env $envs timeout -k 10 300 $B $flags 2>/dev/null | tail -1 | python -c "$P"
  done
done
