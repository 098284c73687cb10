#!/usr/bin/env python3
"""GroupNorm kernel timings on the path's shapes (tuning aid, GPU only)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from adaprompt_amd import ops

dev = torch.device("cuda:0")
from adaprompt_amd import _lib
SHAPES = [(4, 512, 128, torch.float32), (4, 512, 128, torch.bfloat16), (4, 256, 256, torch.float32),
          (4, 128, 512, torch.bfloat16), (4, 64, 320, torch.float32), (4, 64, 320, torch.bfloat16),
          (4, 64, 640, torch.float32), (4, 64, 960, torch.float32), (4, 32, 1920, torch.float32),
          (4, 32, 640, torch.float32), (4, 16, 1280, torch.float32), (4, 8, 2560, torch.float32), (1, 64, 320, torch.float32)]
if len(sys.argv) > 1 and sys.argv[1] == "two_pass":
    os.environ["ADAP_GN_TWO_PASS"] = "1"


def gpu_us(fn, n=20):
    """GPU time per call: ``n`` calls captured into one hipGraph and replayed (eager timing of a 10-20 us kernel from Python
    measures the host: torch.empty x 5 + ctypes is ~15 us per call)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    from adaprompt_amd import ops as _ops
    _ops.set_gn_single_launch_stream(dev, side.cuda_stream)
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(n):
                fn()
        for _ in range(3):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3


for (B, H, C, dt) in SHAPES:
    x = torch.randn(B, H, H, C, device=dev).to(dt)
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    _, y, m, r = ops.groupnorm_fwd(x, g, b, 1e-5, 1)
    dy = torch.randn(B, H, H, C, device=dev).to(torch.bfloat16)
    us = gpu_us(lambda: ops.groupnorm_fwd(x, g, b, 1e-5, 1))
    vf = _lib.call_long("adap_groupnorm_last_variant")
    us2 = gpu_us(lambda: ops.groupnorm_bwd(dy, x, g, b, m, r, 1, out_f32=False, out_bf16=True))
    vb = _lib.call_long("adap_groupnorm_last_variant")
    by = x.numel() * (x.element_size() + 2)                      # algorithmic: x once + y
    by2 = x.numel() * (x.element_size() + 2 + 2)                 # algorithmic: x, dy once + dx
    print(f"B{B} {H}x{H}x{C} {str(dt)[6:]:9s} rows/thread fwd {vf} bwd {vb}  fwd {us:7.1f} us {by / us / 1e3:7.0f} GB/s = "
          f"{by / us / 8e6:.3f} of 8 TB/s   bwd {us2:7.1f} us {by2 / us2 / 1e3:7.0f} GB/s = {by2 / us2 / 8e6:.3f}", flush=True)
