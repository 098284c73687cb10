#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace (tools/trace_run.sh): per stream, the kernel time of the last N steps by kernel family,
launch counts, and the idle time of the stream between consecutive kernels (gap), per family of the FOLLOWING kernel."""
import collections
import csv
import re
import sys

path, nsteps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 6
rows = list(csv.DictReader(open(path)))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])


def fam(n):
    n = re.sub(r"^void ", "", n)
    for k in ("conv3x3_halo", "conv_gemm_ring", "conv_gemm_kernel", "splitk_reduce", "attn_bwd_dkv", "attn_bwd_dq", "attn_fwd", "attn_delta",
              "attn_capture", "attn_tokmap", "gn_fused_fwd", "gn_fused_bwd", "gn_apply", "gn_stats", "gn_", "ln_fwd", "ln_bwd", "geglu", "prodigy",
              "Cijk", "rocclr", "cosine_rows", "hinge", "pack_weight", "gather_rows", "colsum", "concat", "add2", "vae_softmax"):
        if k in n:
            return k
    if "at::native" in n:
        m = re.search(r"(\w+Functor|\w+_kernel_cuda|direct_copy|FillFunctor|reduce_kernel|index\w*|upsample\w*|cat\w*)", n)
        return "aten:" + (m.group(1) if m else "other")
    return n[:36]


# the main leg's timed steps = the last nsteps occurrences of the optimizer... simpler: take the last 40 % of the trace by time
main = [r for r in rows if r["Stream_Id"] == "0"]
# steps are delimited by the masked_mse kernel (once per micro-batch)
marks = [r["s"] for r in main if "masked_mse" in r["Kernel_Name"]]
t_lo, t_hi = marks[-nsteps - 1], marks[-1]
print(f"window: {nsteps} steps, {(t_hi - t_lo) / 1e6 / nsteps:.2f} ms/step wall under the tracer")
for sid in sorted({r["Stream_Id"] for r in rows}):
    ks = sorted([r for r in rows if r["Stream_Id"] == sid and t_lo <= r["s"] < t_hi], key=lambda r: r["s"])
    if not ks:
        continue
    busy = sum(r["e"] - r["s"] for r in ks)
    agg = collections.OrderedDict()
    prev_e = None
    for r in ks:
        d = agg.setdefault(fam(r["Kernel_Name"]), [0, 0, 0])
        d[0] += 1
        d[1] += r["e"] - r["s"]
        if prev_e is not None:
            d[2] += max(0, r["s"] - prev_e)
        prev_e = max(prev_e or 0, r["e"])
    print(f"--- stream {sid}: {len(ks) / nsteps:.0f} launches/step, kernel time {busy / 1e6 / nsteps:.2f} ms/step")
    for k, (c, ns, gap) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"   {k:34s} {c / nsteps:7.1f} /step {ns / 1e6 / nsteps:7.3f} ms  avg {ns / c / 1e3:7.1f} us   gap-before avg {gap / c / 1e3:6.1f} us")
