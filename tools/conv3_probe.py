#!/usr/bin/env python3
"""3x3 stride-1 convolutions of the UNet's ResBlocks as DEPENDENT chains (GroupNorm-free: x -> conv -> conv -> ...), swept
over split-K and the channel tile: what does a launch cost end to end (kernel + split-K reduce + boundaries)?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from adaprompt_amd import _lib, ops

dev = torch.device("cuda:0")


def chain(B, H, C, bn, ks, reps=40):
    x = (torch.randn(B, H, H, C, device=dev) * 0.5).to(torch.bfloat16)
    w = ops.PackedConv(torch.randn(C, C, 3, 3, device=dev) * (9 * C) ** -0.5, torch.zeros(C, device=dev))
    _lib.call("adap_conv2d_debug_force", 0, bn)

    def run(n):
        h = x
        for _ in range(n):
            _, h = ops.conv2d(h, w.fwd, C, 3, 1, 1, bias=w.bias, out_f32=False, out_bf16=True, ksplit=ks)
        return h
    try:
        run(5)
        va = _lib.call_long("adap_conv2d_last_variant")
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            run(reps)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (3 * reps)
    finally:
        _lib.call("adap_conv2d_debug_force", 0, 0)
    return us, va


for (B, H, C) in [(4, 64, 320), (4, 32, 640), (4, 16, 1280), (4, 8, 1280), (2, 64, 320), (2, 32, 640), (2, 16, 1280)]:
    fl = 2.0 * B * H * H * C * C * 9
    print(f"--- B={B} {H}x{H} C={C} ({fl / 1e9:.1f} GFLOP)", flush=True)
    for bn in (0, 128, 160):
        for ks in (0, 1, 2, 3, 4, 6, 8, 12, 16):
            if ks > C // 64:
                continue
            us, va = chain(B, H, C, bn, ks)
            print(f"   bn={bn:3d} ks={ks:2d} (variant {va}): {us:7.1f} us  {fl / us / 1e6:7.1f} TF/s", flush=True)
