#!/usr/bin/env python3
"""one configuration of the 64x64 attention backward, 100 launches (for `rocprofv3 --kernel-trace --stats`):
COUNT_FRAC (empty = no key count), ADAP_ATTN_DKV_QSPLIT."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from adaprompt_amd import ops

dev = torch.device("cuda:0")
B, H, N, d = 4, 8, int(os.environ.get("NQ", "4096")), 40
M = int(os.environ.get("MK", "4096"))
q, do = (torch.randn(B, N, H * d, device=dev).to(torch.bfloat16) for _ in range(2))
k, v = (torch.randn(B, M, H * d, device=dev).to(torch.bfloat16) for _ in range(2))
frac = os.environ.get("COUNT_FRAC", "")
count = torch.full((B,), int(M * float(frac)), device=dev, dtype=torch.int32) if frac else None
o, lse = ops.attention_fwd(q, k, v, H, None, key_count=count)
for _ in range(100):
    ops.attention_bwd(q, k, v, o, do, lse, H, None, key_count=count)
torch.cuda.synchronize()
