#!/bin/bash
# Two independent bench processes on ONE card at the same time (all legs): contention makes streams drift apart and shows
# ordering bugs that a process alone never hits.  Every leg of both lines must be finite.   bash tools/two_bench_soak.sh [pairs]
OUT=gpurun_out; mkdir -p $OUT
for i in $(seq 1 ${1:-2}); do
  ADAP_GN_TWO_PASS=1 timeout -k 10 900 python bench.py --no-cpu-baseline --steps 6 --warmup 2 > $OUT/soak_a$i.log 2>&1 &
  pa=$!
  ADAP_GN_TWO_PASS=1 timeout -k 10 900 python bench.py --no-cpu-baseline --steps 6 --warmup 2 > $OUT/soak_b$i.log 2>&1 &
  pb=$!
  wait $pa; ra=$?; wait $pb; rb=$?
  for f in $OUT/soak_a$i.log $OUT/soak_b$i.log; do
    tail -1 $f | python -c '
import json, sys, math
d = json.loads(sys.stdin.read())
def walk(o, path=""):
    bad = []
    if isinstance(o, dict):
        for k, v in o.items():
            bad += walk(v, path + "/" + k)
    elif isinstance(o, float) and not math.isfinite(o):
        bad.append(path)
    elif o is False and path.endswith("finite"):
        bad.append(path)
    return bad
print(d["value"], d["final_loss"], "NON-FINITE: " + ", ".join(walk(d)) if walk(d) else "all legs finite",
      {k: (v.get("final_loss"), v.get("finite")) for k, v in d.items() if isinstance(v, dict) and ("final_loss" in v or "finite" in v)})'
  done
  echo "exit codes $ra $rb"
done
