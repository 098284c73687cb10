#!/usr/bin/env python3
"""Run a few representative contraction shapes in isolation (for rocprofv3 --pmc passes)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from adaprompt_amd import ops

dev = torch.device("cuda:0")
shapes = [(4, 128, 128, 512, 3), (4, 256, 256, 256, 3), (4, 512, 512, 128, 3), (4, 320, 320, 64, 3), (4, 640, 640, 32, 3)]
for (B, Ci, Co, H, K) in shapes:
    x = torch.randn(B, H, H, Ci, device=dev).to(torch.bfloat16)
    w = torch.randn(Co, Ci, K, K, device=dev) * 0.02
    pk = ops.PackedConv(w, torch.zeros(Co, device=dev))
    for _ in range(3):
        y, _ = ops.conv2d(x, pk.fwd, Co, K, 1, 1, bias=pk.bias)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        y, _ = ops.conv2d(x, pk.fwd, Co, K, 1, 1, bias=pk.bias)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = 2.0 * B * H * H * Co * Ci * K * K
    print(f"B{B} {Ci}->{Co} @{H} k{K}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TF/s  x {x.numel() * 2 / 1e6:.0f} MB  y {y.numel() * 4 / 1e6:.0f} MB")
