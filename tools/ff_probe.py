#!/usr/bin/env python3
"""The feed-forward's four contractions at the UNet's levels as the step issues them (fused GEGLU epilogues, bf16 operands):
FF1 forward (C -> 8C, GEGLU), FF2 forward (4C -> C, + residual), FF2 data gradient (C -> 4C, GEGLU'), FF1 data gradient (8C -> C),
each timed back to back under sustained load, per kernel variant (adap_conv2d_debug_force)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from adaprompt_amd import _lib, ops

dev = torch.device("cuda:0")
BF16 = torch.bfloat16
KIND = {0: "auto", 1: "gemm", 2: "ring256", 3: "ring128"}
ITERS = int(os.environ.get("ITERS", "100"))


def timed(fn):
    for _ in range(ITERS // 2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(ITERS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / ITERS * 1e3


for rows, C in ((16384, 320), (4096, 640), (1024, 1280)):
    x = (torch.randn(rows, C, device=dev) * 0.5).to(BF16)
    ff1 = ops.PackedConv(torch.randn(8 * C, C, device=dev) * C ** -0.5, torch.zeros(8 * C, device=dev))
    ff2 = ops.PackedConv(torch.randn(C, 4 * C, device=dev) * (4 * C) ** -0.5, torch.zeros(C, device=dev))
    res = torch.randn(rows, C, device=dev)
    g16 = (torch.randn(rows, C, device=dev) * 0.1).to(BF16)
    hh, gg = ops.linear_geglu_fwd(x, ff1)
    dh = ops.linear_geglu_bwd(g16, ff2, hh)
    print(f"--- rows {rows} C {C}", flush=True)
    for kind, bn in ((0, 0), (1, 128), (1, 64), (2, 128), (3, 128), (3, 64), (1, 160), (2, 160), (3, 160)):
        _lib.call("adap_conv2d_debug_force", kind, bn)
        try:
            t1 = timed(lambda: ops.linear_geglu_fwd(x, ff1))
            v1 = _lib.call_long("adap_conv2d_last_variant")
            t2 = timed(lambda: ops.linear(gg, ff2.fwd, C, bias=ff2.bias, residual=res, out_f32=False, out_bf16=True))
            v2 = _lib.call_long("adap_conv2d_last_variant")
            t3 = timed(lambda: ops.linear_geglu_bwd(g16, ff2, hh))
            v3 = _lib.call_long("adap_conv2d_last_variant")
            t4 = timed(lambda: ops.linear(dh, ff1.bwd, C, out_f32=True, out_bf16=False))
            v4 = _lib.call_long("adap_conv2d_last_variant")
        finally:
            _lib.call("adap_conv2d_debug_force", 0, 0)
        f1 = 2.0 * rows * 8 * C * C
        print(f"  {KIND[kind]:8s} bn={bn:3d}: FF1 {t1:6.1f} us ({f1 / t1 / 1e6:5.0f} TF/s, v{v1})  FF2 {t2:6.1f} (v{v2})  FF2' {t3:6.1f} (v{v3})  "
              f"FF1' {t4:6.1f} (v{v4})", flush=True)
