#!/usr/bin/env python3
"""The attention backward (dQ + dK/dV kernels) under sustained load on the UNet's self-attention shapes: 150 warm-up + 150 timed
calls each (ADAP_LIB_PATH selects another build for A/B runs)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from adaprompt_amd import ops

dev = torch.device("cuda:0")
for (B, H, N, M, d) in [(4, 8, 4096, 4096, 40), (4, 8, 1024, 1024, 80), (4, 8, 4096, 77, 40)]:
    C = H * d
    q, k, v, do = (torch.randn(B, n, C, device=dev).to(torch.bfloat16) for n in (N, M, M, N))
    o, lse = ops.attention_fwd(q, k, v, H)
    for _ in range(150):
        ops.attention_bwd(q, k, v, o, do, lse, H)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(150):
        ops.attention_bwd(q, k, v, o, do, lse, H)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 150 * 1e3
    print(f"N={N} M={M} d={d}: bwd {us:7.1f} us  {2.5 * 4.0 * B * H * N * M * d / us / 1e6:6.1f} TF/s", flush=True)
