#!/usr/bin/env python3
"""The UNet's attention shapes with q | k | v as slices of ONE fused [B, N, 3C] buffer (how the step lays them out), forward and
backward, timed under sustained load (ITERS warm-up + ITERS timed).  ADAP_ATTN_XCD=0 / 1 selects the plain / XCD-aware
workgroup map (attention.hip attn_wg_coords); run it twice for the A/B, and under ``rocprofv3 --pmc FETCH_SIZE --kernel-trace``
with ITERS=5 for the per-launch traffic (tools/pmc_summary.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from adaprompt_amd import ops

dev = torch.device("cuda:0")
ITERS = int(os.environ.get("ITERS", "150"))
B = int(os.environ.get("BATCH", "4"))
SHAPES = [(8, 4096, 4096, 40, "self 64x64"), (8, 1024, 1024, 80, "self 32x32"), (8, 256, 256, 160, "self 16x16"),
          (8, 64, 64, 160, "self 8x8"), (8, 4096, 77, 40, "cross 64x64"), (8, 1024, 77, 80, "cross 32x32"), (8, 256, 77, 160, "cross 16x16")]


def timed(fn):
    for _ in range(ITERS):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(ITERS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / ITERS * 1e3


print(f"ADAP_ATTN_XCD={os.environ.get('ADAP_ATTN_XCD', '1')} B={B} iters={ITERS}", flush=True)
for H, N, M, d, name in SHAPES:
    C = H * d
    if M == N:
        qkv = torch.randn(B, N, 3 * C, device=dev).to(torch.bfloat16)
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    else:
        q = torch.randn(B, N, C, device=dev).to(torch.bfloat16)
        kv = torch.randn(B, M, 2 * C, device=dev).to(torch.bfloat16)
        k, v = kv[..., :C], kv[..., C:]
    do = torch.randn(B, N, C, device=dev).to(torch.bfloat16)
    o, lse = ops.attention_fwd(q, k, v, H)
    tf = timed(lambda: ops.attention_fwd(q, k, v, H))
    tb = timed(lambda: ops.attention_bwd(q, k, v, o, do, lse, H))
    fl = 4.0 * B * H * N * M * d
    print(f"{name:12s} N={N:5d} M={M:5d} d={d:3d}: fwd {tf:7.1f} us {fl / tf / 1e6:6.1f} TF/s   bwd {tb:7.1f} us {2.5 * fl / tb / 1e6:6.1f} TF/s",
          flush=True)
