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
for (B, H, C, dt) in SHAPES:
    x = torch.randn(B, H, H, C, device=dev).to(dt)
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    for _ in range(3):
        ops.groupnorm_fwd(x, g, b, 1e-5, 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        _, y, m, r = ops.groupnorm_fwd(x, g, b, 1e-5, 1)
    e1.record()
    vf = _lib.call_long("adap_groupnorm_last_variant")
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    by = x.numel() * (x.element_size() + 2)                      # algorithmic: x once + y
    dy = torch.randn(B, H, H, C, device=dev).to(torch.bfloat16)
    e0.record()
    for _ in range(10):
        ops.groupnorm_bwd(dy, x, g, b, m, r, 1, out_f32=False, out_bf16=True)
    e1.record()
    torch.cuda.synchronize()
    ms2 = e0.elapsed_time(e1) / 10
    vb = _lib.call_long("adap_groupnorm_last_variant")
    by2 = x.numel() * (x.element_size() + 2 + 2)                 # algorithmic: x, dy once + dx
    print(f"B{B} {H}x{H}x{C} {str(dt)[6:]:9s} rows/thread fwd {vf} bwd {vb}  fwd {ms * 1e3:7.1f} us {by / ms / 1e6:7.0f} GB/s(algorithmic)   "
          f"bwd {ms2 * 1e3:7.1f} us {by2 / ms2 / 1e6:7.0f} GB/s")
