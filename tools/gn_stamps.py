#!/usr/bin/env python3
"""Where a single-launch GroupNorm forward spends its time: s_memtime stamps per workgroup (diagnostic build path, off in
normal runs: the kernel checks one word of its sync buffer)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from adaprompt_amd import _lib, ops

dev = torch.device("cuda:0")
for (B, H, C) in [(4, 64, 320), (4, 16, 1280), (4, 64, 960)]:
    x = torch.randn(B, H, H, C, device=dev)
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    for _ in range(5):
        ops.groupnorm_fwd(x, g, b, 1e-5, 1)
    torch.cuda.synchronize()
    sync = ops._GN_SYNC[dev.index]
    HDR, WGS = 64, 512
    st0 = HDR + WGS * 128
    sync[3] = 1
    res = []
    for _ in range(20):
        ops.groupnorm_fwd(x, g, b, 1e-5, 1)
        torch.cuda.synchronize()
        st = sync[st0:st0 + WGS * 16].view(torch.int64).view(WGS, 8).cpu()
        n = int(_lib.call_long("adap_groupnorm_last_variant") and (st[:, 0] != 0).sum())
        st = st[:n].double()
        res.append(st[:, 1:6] - st[:, :1])            # per workgroup, relative to its OWN start (s_memtime is per XCD)
        sync[st0:st0 + WGS * 16] = 0
    sync[3] = 0
    r = torch.stack(res).median(0).values            # [wgs, 5] shader-clock ticks
    names = ["loaded+reduced", "swept", "stores issued", "closed", "statistics done"]
    print(f"B{B} {H}x{H}x{C}: {r.shape[0]} workgroups; ticks since the workgroup's own start: median / p90 over workgroups")
    for k in (0, 1, 4, 2, 3):
        col = r[:, k].sort().values
        print(f"   {names[k]:16s} median {float(col[len(col) // 2]):8.0f}  p90 {float(col[int(len(col) * 0.9)]):8.0f}")
