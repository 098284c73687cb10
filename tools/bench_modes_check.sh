COMMON="--steps 6 --warmup 3 --no-cpu-baseline --no-distill-mix --no-ddim --no-unfrozen --no-compos --no-zs-frontend --no-roofline --no-rehearse-exchange --no-entry-leg"
for m in "--entry lightning" "--no-lanes" "--fuse" "--graph" "--no-prefetch" "--batch 2"; do
  echo -n "$m: "; ADAP_BENCH_WATCHDOG=200 timeout -k 10 300 python bench.py $COMMON $m 2>/tmp/err.log | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["final_loss"], d["micro_batch_lanes"], d["hipgraph"], d.get("entry"))' || tail -5 /tmp/err.log
done
