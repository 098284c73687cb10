#!/usr/bin/env python3
"""Phase timing of the ping-pong attention forward (workgroup (0,0), waves 0 and 4): shader-clock ticks per phase."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from adaprompt_amd import _lib, ops

dev = torch.device("cuda:0")
B, H, N, M, d = 4, 8, 4096, 4096, 40
C = H * d
q, k, v = (torch.randn(B, n, C, device=dev).to(torch.bfloat16) for n in (N, M, M))
for _ in range(3):
    ops.attention_fwd(q, k, v, H)
nph = 2 * (M // 64) + 1
buf = torch.zeros(nph * 2 * 3, device=dev, dtype=torch.int64)
_lib.call("adap_attention_set_stamp_buffer", buf.data_ptr())
ops.attention_fwd(q, k, v, H)
torch.cuda.synchronize()
_lib.call("adap_attention_set_stamp_buffer", 0)
st = buf.view(nph, 2, 3).cpu().double()
for g in (0, 1):
    work = (st[4:-4, g, 1] - st[4:-4, g, 0])
    wait = (st[4:-4, g, 2] - st[4:-4, g, 1])
    even, odd = work[0::2], work[1::2]
    print(f"group {g}: work ticks by phase parity: even {float(even.median()):.0f} odd {float(odd.median()):.0f}; barrier wait even "
          f"{float(wait[0::2].median()):.0f} odd {float(wait[1::2].median()):.0f}")
tot = float(st[-1, 1, 2] - st[0, 0, 0])
print(f"whole loop {tot:.0f} ticks = {tot / nph:.0f} per phase")
