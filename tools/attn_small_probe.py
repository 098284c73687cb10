#!/usr/bin/env python3
"""The attention launches of the UNet's 32 x 32, 16 x 16 and 8 x 8 levels (self and cross) timed back to back on hot caches --
to set against their rocprofv3 averages inside the training step (cold operands)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from adaprompt_amd import ops

dev = torch.device("cuda:0")
B, heads = 4, 8


def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for (hw, C) in [(32, 640), (16, 1280), (8, 1280)]:
    N = hw * hw
    for M in (N, 77):
        q = torch.randn(B, N, C, device=dev).to(torch.bfloat16)
        k = torch.randn(B, M, C, device=dev).to(torch.bfloat16)
        v = torch.randn(B, M, C, device=dev).to(torch.bfloat16)
        do = torch.randn(B, N, C, device=dev).to(torch.bfloat16)
        out, lse = ops.attention_fwd(q, k, v, heads)
        f = timed(lambda: ops.attention_fwd(q, k, v, heads))
        b = timed(lambda: ops.attention_bwd(q, k, v, out, do, lse, heads))
        fl = 4.0 * B * heads * N * M * (C // heads)
        print(f"N={N:5d} M={M:5d} d={C // heads:3d}: forward {f:6.1f} us ({fl / f / 1e6:6.1f} TF/s), backward (dq + dk/dv) {b:6.1f} us", flush=True)
