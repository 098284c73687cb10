#!/usr/bin/env python3
"""Latency of the UNet's 1x1 / Linear contractions as DEPENDENT chains (x -> y -> x -> ...: what the training step issues),
swept over kernel variant, channel tile and split-K (adap_conv2d_debug_force + the ksplit argument)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from adaprompt_amd import _lib, ops

dev = torch.device("cuda:0")
KIND = {0: "auto", 1: "gemm", 2: "ring256", 3: "ring128"}


def chain(M, N, K, kind, bn, ks, reps=60):
    """alternate [M,K] -> [M,N] -> [M,K] ...; returns us per launch (mean over both directions)"""
    x = (torch.randn(1, M, 1, K, device=dev) * 0.5).to(torch.bfloat16)
    wa = ops.PackedConv(torch.randn(N, K, device=dev) * K ** -0.5, torch.zeros(N, device=dev))
    wb = ops.PackedConv(torch.randn(K, N, device=dev) * N ** -0.5, torch.zeros(K, device=dev))
    res_a = torch.randn(1, M, 1, N, device=dev)
    res_b = torch.randn(1, M, 1, K, device=dev)
    _lib.call("adap_conv2d_debug_force", kind, bn)

    def run(n):
        h = x
        for _ in range(n):
            _, y = ops.conv2d(h, wa.fwd, N, 1, bias=wa.bias, residual=res_a, out_f32=True, out_bf16=True, ksplit=ks)
            _, h = ops.conv2d(y, wb.fwd, K, 1, bias=wb.bias, residual=res_b, out_f32=True, out_bf16=True, ksplit=ks)
        return h
    try:
        run(10)
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
        us = e0.elapsed_time(e1) * 1e3 / (3 * 2 * reps)
    finally:
        _lib.call("adap_conv2d_debug_force", 0, 0)
    return us, va


shapes = [(16384, 320, 320), (4096, 640, 640), (1024, 1280, 1280), (16384, 2560, 320), (16384, 1280, 320), (4096, 5120, 640),
          (1024, 10240, 1280), (256, 1280, 1280)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for (M, N, K) in shapes:
    fl = 2.0 * M * N * K
    print(f"--- M={M} N={N} K={K}  ({fl / 1e9:.2f} GFLOP; pair = [M,K]->[M,N]->[M,K])", flush=True)
    base, va = chain(M, N, K, 0, 0, 0)
    print(f"   auto (variant {va}, library split-K plan): {base:7.1f} us  {fl / base / 1e6:7.1f} TF/s", flush=True)
    for kind in (1, 2, 3):
        for bn in (64, 128, 160):
            if kind == 2 and bn == 64:
                continue
            for ks in (1, 2, 4):
                if K // 64 < 4 * ks:
                    continue
                us, va = chain(M, N, K, kind, bn, ks)
                print(f"   {KIND[kind]:8s} bn={bn:3d} ks={ks}: {us:7.1f} us  {fl / us / 1e6:7.1f} TF/s", flush=True)
