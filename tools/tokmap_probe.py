#!/usr/bin/env python3
"""The token-map family in the shapes the training step runs it (12 capture layers, bs 4): forward (attention_capture, maps
only) and the backward's prep (kw, gq, reduce), each timed back to back and alone."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from adaprompt_amd import _lib, ops

dev = torch.device("cuda:0")
B, M, G = 4, 77, 2


def timed(fn, n=20):
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


for (hw, C, heads, layers) in [(64, 320, 8, 3), (32, 640, 8, 3), (16, 1280, 8, 5), (8, 1280, 8, 1)]:
    N, d = hw * hw, C // heads
    # q2 / kv2 as the block holds them: q [B, N, C] bf16, k a column slice of kv2 [B, M, 2C]
    q = torch.randn(B, N, C, device=dev).to(torch.bfloat16)
    kv = torch.randn(B, M, 2 * C, device=dev).to(torch.bfloat16)
    k = kv[..., :C]
    w = torch.zeros(B, M, G, device=dev)
    w[:, 4:20, 0] = 1.0
    w[:, 24:28, 1] = 1.0
    dT = torch.randn(B, heads, N, G, device=dev)
    fwd = timed(lambda: ops.attention_capture(q, k, heads, tok_w=w, dense=False))
    prep = timed(lambda: ops.attention_tokmap_prep(dT, w, q, k, heads))
    print(f"{hw}x{hw} C={C} d={d} ({layers} layers): forward {fwd:6.1f} us, backward prep (kw + gq + reduce) {prep:6.1f} us;"
          f"  q {q.numel() * 2 / 1e6:.1f} MB", flush=True)
