#!/usr/bin/env python3
"""The VAE encoder's conv_in (3 -> 128 @ 512 x 512, bs 4): the single-K-step kernel against the general path (pad to 8 channels +
nine K steps), back-to-back launches timed with HIP events."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from adaprompt_amd import ops

dev = torch.device("cuda:0")
B, H, W, Cout = 4, 512, 512, 128
x = torch.randn(B, H, W, 3, device=dev)
pk = ops.PackedConv(torch.randn(Cout, 3, 3, 3, device=dev) * 27 ** -0.5, torch.randn(Cout, device=dev))


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


a = timed(lambda: ops.conv3x3_rgb(x, pk.fwd, Cout, bias=pk.bias, gn_stats=True))
b = timed(lambda: ops.conv2d(ops.pad_cast_bf16(x, 8), pk.fwd, Cout, 3, 1, 1, bias=pk.bias, gn_stats=True))
a2 = timed(lambda: ops.conv3x3_rgb(x, pk.fwd, Cout, bias=pk.bias, gn_stats=False))
b2 = timed(lambda: ops.conv2d(ops.pad_cast_bf16(x, 8), pk.fwd, Cout, 3, 1, 1, bias=pk.bias, gn_stats=False))
y = torch.empty(B, H, W, Cout, device=dev)
c2 = timed(lambda: y.fill_(1.0))
print(f"without the statistics epilogue: single K step {a2:.1f} us, general {b2:.1f} us; torch fill of the same 537 MB: {c2:.1f} us")
gb = B * H * W * (Cout * 4 + 12) / 1e9
print(f"conv_in 3 -> {Cout} @ {H}x{W} bs {B}: single K step {a:.1f} us ({gb / a * 1e6:.0f} GB/s of output + input)   "
      f"general path (pad_cast + 9 K steps) {b:.1f} us")
