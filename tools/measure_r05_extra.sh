#!/bin/bash
# round 5's extra measurements (run through gpurun from the repo root, after tools/measure_round.sh r05):
#   attention per-launch HBM traffic with the plain / XCD-aware workgroup map (two --pmc FETCH_SIZE passes of tools/attn_xcd_probe.py)
#   and the same probe's sustained timings; the feed-forward contractions with narrow / regrouped bf16 stores; the VAE alone
set -e -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
echo "[extra] attention timings"; date
for m in 0 1 7 15; do ADAP_ATTN_XCD=$m ITERS=200 timeout -k 10 300 python tools/attn_xcd_probe.py 2>/dev/null; done > "$OUT/r05_attn_xcd_timings.log"
cat "$OUT/r05_attn_xcd_timings.log"
echo "[extra] attention FETCH_SIZE"; date
cd /tmp && export TMPDIR=/tmp
for m in 0 7 15; do
  ADAP_ATTN_XCD=$m ITERS=5 timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d "$OUT/r05_attn_pmcF_xcd$m" -o x -- python3 "$ROOT/tools/attn_xcd_probe.py" > "$OUT/r05_attn_pmcF_xcd$m.log" 2>&1
  find "$OUT/r05_attn_pmcF_xcd$m" -name "*kernel_trace.csv" -delete || true
done
cd "$ROOT"
python3 - <<'PY'
import csv, glob, json, sys
csv.field_size_limit(sys.maxsize)
res = {}
for tag in ("xcd0", "xcd7", "xcd15"):
    f = glob.glob(f"gpurun_out/r05_attn_pmcF_{tag}/**/*counter_collection.csv", recursive=True)[0]
    per = {}
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] != "FETCH_SIZE" or "attn_" not in row["Kernel_Name"]:
            continue
        k = row["Kernel_Name"].split("(")[0].replace("void ", "") + " grid=" + row["Grid_Size"]
        e = per.setdefault(k, [0, 0.0])
        e[0] += 1
        e[1] += float(row["Counter_Value"])
    res[tag] = {k: {"launches": n, "fetch_MB_per_launch": round(2 * kb / n / 1e3, 1)} for k, (n, kb) in sorted(per.items())}
json.dump({"what": "HBM fetch per attention launch, tools/attn_xcd_probe.py shapes (B4 h8; q|k|v slices of one fused buffer), "
                   "FETCH_SIZE (KB) doubled per the gfx950 correction; xcd0 = plain blockIdx map, xcd7 = the shipped rule "
                   "(forward always, backward kernels at <= 16 blocks per (batch, head) row), xcd15 = the map forced everywhere",
           "by_map": res}, open("gpurun_out/r05_attn_traffic.json", "w"), indent=1)
for tag, d in res.items():
    for k, v in d.items():
        if "131072" in k or "262144" in k:
            print(tag, k, v)
PY
echo "[extra] feed-forward contractions, regrouped / narrow bf16 stores"; date
(ITERS=100 timeout -k 10 300 python tools/ff_probe.py 2>/dev/null | grep -E "^---|auto"; echo "ADAP_CONV_DEBUG=8 (narrow stores)"; ADAP_CONV_DEBUG=8 ITERS=100 timeout -k 10 300 python tools/ff_probe.py 2>/dev/null | grep -E "^---|auto") | tee "$OUT/r05_ff_probe.log"
echo "[extra] VAE alone"; date
(for v in 1 0; do echo "ADAP_VAE_DOWNSAMPLE_BF16=$v $(ADAP_VAE_DOWNSAMPLE_BF16=$v timeout -k 10 300 python tools/vae_probe.py 2>&1 | tail -1)"; done) | tee "$OUT/r05_vae_probe.log"
echo "[extra] done"; date
