#!/bin/bash
set -e -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
echo "[r05d] VAE alone"; date
timeout -k 10 300 python tools/vae_probe.py 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
ITERS=10 timeout -k 10 300 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/r05d_vae_stats" -o vae -- python3 "$ROOT/tools/vae_probe.py" > "$OUT/r05d_vae_stats.log" 2>&1
find "$OUT/r05d_vae_stats" -name "*kernel_trace.csv" -exec cp {} "$OUT/r05d_vae_kernel_trace.csv" \;
find "$OUT/r05d_vae_stats" -name "*kernel_trace.csv" -delete || true
cd "$ROOT"
python - <<'PY'
import csv,sys,collections
csv.field_size_limit(sys.maxsize)
rows=list(csv.DictReader(open('gpurun_out/r05d_vae_kernel_trace.csv')))
# the last call's launches: take the final 1/15th of the trace (5 warm-up + 10 timed calls)
n=len(rows)//15
last=rows[-n:]
tot=0
agg=collections.OrderedDict()
for r in last:
    k=(r['Kernel_Name'].split('(')[0][:70], r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size',''))
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    e=agg.setdefault(k,[0,0.0]); e[0]+=1; e[1]+=d; tot+=d
for k,(c,t) in sorted(agg.items(), key=lambda kv:-kv[1][1]):
    print(f"{k[0]:72s} grid {k[1]:>8s} x{c:3d} {t:9.1f} us")
print("launches", n, "total us", round(tot,1))
PY
echo "[r05d] compaction threshold A/B"; date
COMMON="--steps 20 --warmup 5 --no-cpu-baseline --no-distill-mix --no-ddim --no-unfrozen --no-compos --no-zs-frontend --no-roofline --no-rehearse-exchange --no-entry-leg"
for rep in 1 2; do
  for n in 1024 4096; do
    ADAP_COMPACT_KEYS_MIN_N=$n timeout -k 10 600 python bench.py $COMMON > "$OUT/r05d_bench_c${n}_$rep.log" 2>&1
    echo "compact_min_n=$n rep=$rep $(tail -1 $OUT/r05d_bench_c${n}_$rep.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["host_work_ms_per_step"])')"
  done
done
rm -f "$OUT/r05d_vae_kernel_trace.csv"
echo "[r05d] done"; date
