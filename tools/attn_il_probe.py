#!/usr/bin/env python3
"""The forward attention kernels side by side on the 64x64 self-attention (B4 h8 N4096 d40): correctness against an f32 torch
softmax(QK^T)V and time under sustained load (300 warm-up + 300 timed launches each).  Modes (adap_attention_set_debug's first
argument): 0 default choice, 5 attn_fwd_kernel, 3 / 4 interleaved with 2 / 1 query blocks per wave, 2 ping-pong."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from adaprompt_amd import _lib, ops

dev = torch.device("cuda:0")
B, H, N, d = 4, 8, int(os.environ.get("N", 4096)), 40
torch.manual_seed(0)
q, k, v = (torch.randn(B, N, H * d, device=dev).to(torch.bfloat16) for _ in range(3))
q = q * 1.5


def ref(q, k, v, km):
    qf, kf, vf = (t.float().view(B, N, H, d).permute(0, 2, 1, 3) for t in (q, k, v))
    out = torch.empty(B, H, N, d, device=dev)
    for b in range(B):
        s = qf[b] @ kf[b].transpose(1, 2) * d ** -0.5
        if km is not None:
            s = s.masked_fill(km[b].view(1, 1, N) == 0, -torch.finfo(torch.float32).max)
        out[b] = torch.softmax(s, -1) @ vf[b]
    return out.permute(0, 2, 1, 3).reshape(B, N, H * d)


m2 = torch.zeros(B, 64, N // 64, dtype=torch.uint8, device=dev)
m2[:, 5:59, 5:(N // 64 - 5)] = 1
masks = {"nomask": None, "mask": m2.view(B, N).contiguous()}
modes = [int(x) for x in os.environ.get("MODES", "5,3,4,0").split(",")]
for name, km in masks.items():
    want = ref(q, k, v, km)
    for mode in modes:
        _lib.call("adap_attention_set_debug", mode, -1, int(os.environ.get("QB1", "0")), -1)
        o, lse = ops.attention_fwd(q, k, v, H, km)
        var = _lib.call_long("adap_attention_fwd_last_variant")
        err = float((o.float() - want).abs().max()) / float(want.abs().max())
        for _ in range(300):
            ops.attention_fwd(q, k, v, H, km)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300):
            ops.attention_fwd(q, k, v, H, km)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 300 * 1e3
        print(f"{name} mode {mode} variant {var}: max err {err:.2e}   {us:.1f} us  {4.0 * B * H * N * N * d / us / 1e6:.1f} TF/s", flush=True)
_lib.call("adap_attention_set_debug", 0, -1, -1, -1)
