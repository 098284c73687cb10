#!/usr/bin/env python3
"""Self-attention kernels in isolation on the UNet's shapes (tuning aid, GPU only).  Self-attention shapes (N == M) run in the form
the training step calls: pre-scaled queries, scale = 0 (functional.PRESCALE_Q); the cross-attention shape with the kernel's own scale."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from adaprompt_amd import ops

dev = torch.device("cuda:0")
for (B, H, N, M, d) in [(4, 8, 4096, 4096, 40), (4, 8, 1024, 1024, 80), (4, 8, 256, 256, 160), (4, 8, 4096, 77, 40)]:
    C = H * d
    q = torch.randn(B, N, C, device=dev).to(torch.bfloat16)
    k = torch.randn(B, M, C, device=dev).to(torch.bfloat16)
    v = torch.randn(B, M, C, device=dev).to(torch.bfloat16)
    do = torch.randn(B, N, C, device=dev).to(torch.bfloat16)
    sc = 0.0 if N == M else None
    for _ in range(2):
        o, lse = ops.attention_fwd(q, k, v, H, scale=sc)
        ops.attention_bwd(q, k, v, o, do, lse, H, scale=sc)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    for _ in range(5):
        o, lse = ops.attention_fwd(q, k, v, H, scale=sc)
    e[1].record()
    for _ in range(5):
        ops.attention_bwd(q, k, v, o, do, lse, H, scale=sc)
    e[2].record()
    torch.cuda.synchronize()
    f = 4.0 * B * H * N * M * d
    t1, t2 = e[0].elapsed_time(e[1]) / 5, e[1].elapsed_time(e[2]) / 5
    from adaprompt_amd import _lib
    print(f"[fwd variant {_lib.call_long('adap_attention_fwd_last_variant')}] N={N} M={M} d={d}: fwd {t1 * 1e3:7.1f} us {f / t1 / 1e9:6.1f} TF/s   bwd {t2 * 1e3:7.1f} us {2.5 * f / t2 / 1e9:6.1f} TF/s (2.5x fwd flops)")
