#!/usr/bin/env python3
"""attention backward at the 64x64 level with and without a per-sample key count (sustained: 200 warm-up + 200 timed)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from adaprompt_amd import ops

dev = torch.device("cuda:0")
B, H, N, d = 4, 8, 4096, 40
q, k, v, do = (torch.randn(B, N, H * d, device=dev).to(torch.bfloat16) for _ in range(4))


def run(count):
    o, lse = ops.attention_fwd(q, k, v, H, None, key_count=count)
    for _ in range(200):
        ops.attention_bwd(q, k, v, o, do, lse, H, None, key_count=count)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        ops.attention_bwd(q, k, v, o, do, lse, H, None, key_count=count)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 200 * 1e3


print("ADAP_ATTN_DKV_QSPLIT =", os.environ.get("ADAP_ATTN_DKV_QSPLIT", "(heuristic)"))
print("no count            :", round(run(None), 1), "us (dq + dkv + delta)")
for frac in (0.72, 0.5, 0.04, 0.0):
    c = torch.full((B,), int(N * frac), device=dev, dtype=torch.int32)
    print(f"count = {frac:4.2f} N      :", round(run(c), 1), "us")
