#!/usr/bin/env python3
"""time adap_attention_capture at the three distillation-layer shapes (B=4, 8 heads, M=77)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaprompt_amd import ops
dev = torch.device("cuda:0")
for N, d in ((64, 160), (256, 160), (1024, 80), (4096, 40)):
    q = torch.randn(4, N, 8 * d, device=dev).bfloat16()
    k = torch.randn(4, 77, 8 * d, device=dev).bfloat16()
    for _ in range(3):
        ops.attention_capture(q, k, 8)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.attention_capture(q, k, 8)
    e1.record()
    torch.cuda.synchronize()
    print(f"N={N} d={d}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
