#!/usr/bin/env python3
"""In-kernel clock of the dominant conv kernel (MI355X_MICROARCH.md 'DVFS give-back', item 6): run the stencil-window
kernel back to back on random data for > 2 s, then stamp one launch: clock = d(s_memtime) / d(s_memrealtime) * 100 MHz,
median over workgroups.  Prints the MFMA-bound time of the launch at that clock next to the measured time."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from adaprompt_amd import _lib, ops

dev = torch.device("cuda:0")
for (B, C, H) in [(4, 512, 128), (4, 256, 256), (4, 128, 512), (4, 1280, 16)]:
    x = torch.randn(B, H, H, C, device=dev).to(torch.bfloat16)
    w = torch.randn(C, C, 3, 3, device=dev) * 0.02
    pk = ops.PackedConv(w, torch.zeros(C, device=dev))
    flops = 2.0 * B * H * H * C * C * 9
    ops.conv2d(x, pk.fwd, C, 3, 1, 1, bias=pk.bias)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.conv2d(x, pk.fwd, C, 3, 1, 1, bias=pk.bias)
    e1.record()
    torch.cuda.synchronize()
    one = e0.elapsed_time(e1) * 1e-3
    n = max(10, int(2.2 / one))
    for _ in range(n):                                     # > 2 s of back-to-back launches: the clock has settled
        ops.conv2d(x, pk.fwd, C, 3, 1, 1, bias=pk.bias)
    nwg = B * (H * H // 256) * ((C + 127) // 128) * 16    # upper bound incl. split-K slices
    buf = torch.zeros(2 * nwg, device=dev, dtype=torch.int64)
    _lib.call("adap_conv2d_set_clock_probe", buf.data_ptr())
    e0.record()
    ops.conv2d(x, pk.fwd, C, 3, 1, 1, bias=pk.bias)
    e1.record()
    _lib.call("adap_conv2d_set_clock_probe", 0)
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    v = buf.view(-1, 2).cpu()
    v = v[v[:, 1] > 0].double()
    ghz = (v[:, 0] / v[:, 1] * 0.1)
    clk = float(ghz.median())
    peak_at_clk = 2.5e15 * clk / 2.4                        # nominal 2.5 PF is at 2.4 GHz
    print(f"B{B} {C}->{C} @{H}: {us:7.1f} us = {flops / us / 1e6:7.1f} TF/s | in-kernel clock {clk:.2f} GHz "
          f"(p10 {float(ghz.quantile(0.1)):.2f}, p90 {float(ghz.quantile(0.9)):.2f}; {len(v)} workgroups, loop "
          f"{float((v[:, 1] * 10).median()) / 1e3:.1f} us median) -> MFMA peak at that clock {peak_at_clk / 1e12:.0f} TF/s, "
          f"kernel at {flops / (us * 1e-6) / peak_at_clk:.1%} of it")
