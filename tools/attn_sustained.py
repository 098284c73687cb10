#!/usr/bin/env python3
"""The 64x64 self-attention forward under sustained load (the clock the chip holds in the training step, not the first
launches' boost clock): 400 warm-up launches, then 400 timed.  ADAP_ATTN_FORCE_PP=1 selects the ping-pong kernel, ADAP_ATTN_PP_PRIO its priority mode."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from adaprompt_amd import _lib, ops

dev = torch.device("cuda:0")
B, H, N, d = 4, 8, 4096, 40
q, k, v = (torch.randn(B, N, H * d, device=dev).to(torch.bfloat16) for _ in range(3))
km = None
if os.environ.get("MASKED"):          # the recon iteration's case: a key mask with a masked border (img_mask)
    m2 = torch.zeros(B, 64, 64, dtype=torch.uint8, device=dev)
    m2[:, 5:59, 5:59] = 1
    km = m2.view(B, N).contiguous()
for _ in range(400):
    ops.attention_fwd(q, k, v, H, km)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(400):
    ops.attention_fwd(q, k, v, H, km)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 400 * 1e3
print(f"masked={int(km is not None)} prio={os.environ.get('ADAP_ATTN_PP_PRIO', '1')} force_pp={os.environ.get('ADAP_ATTN_FORCE_PP', '0')} variant "
      f"{_lib.call_long('adap_attention_fwd_last_variant')}: {us:.1f} us  {4.0 * B * H * N * N * d / us / 1e6:.1f} TF/s", flush=True)
