#!/usr/bin/env python3
"""In-kernel timeline of a DEPENDENT chain of 1x1 / Linear contractions (the ring kernel's diagnostic stamps, 100 MHz real-time
clock): per launch, when its workgroups start, when their first operand tile has landed, when the K loop ends and when the
stores have drained -- and the gap between one launch's last store and the next launch's first workgroup."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from adaprompt_amd import _lib, ops

dev = torch.device("cuda:0")


def timeline(M, N, K, kind, bn, ks, n=24, graph=True):
    x = (torch.randn(1, M, 1, K, device=dev) * 0.5).to(torch.bfloat16)
    wa = ops.PackedConv(torch.randn(N, K, device=dev) * K ** -0.5, torch.zeros(N, device=dev))
    wb = ops.PackedConv(torch.randn(K, N, device=dev) * N ** -0.5, torch.zeros(K, device=dev))
    res_a = torch.randn(1, M, 1, N, device=dev)
    res_b = torch.randn(1, M, 1, K, device=dev)
    WG = 4096
    clk = torch.zeros(2 * n, WG, 4, device=dev, dtype=torch.int64)
    _lib.call("adap_conv2d_debug_force", kind, bn)

    def run(stamped):
        h = x
        for i in range(n):
            if stamped:
                _lib.call("adap_conv2d_set_clock_probe", clk[2 * i].data_ptr())
            _, y = ops.conv2d(h, wa.fwd, N, 1, bias=wa.bias, residual=res_a, out_f32=True, out_bf16=True, ksplit=ks)
            if stamped:
                _lib.call("adap_conv2d_set_clock_probe", clk[2 * i + 1].data_ptr())
            _, h = ops.conv2d(y, wb.fwd, K, 1, bias=wb.bias, residual=res_b, out_f32=True, out_bf16=True, ksplit=ks)
        return h
    try:
        run(False)
        torch.cuda.synchronize()
        if graph:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                run(True)
            g.replay()
            clk.zero_()
            g.replay()
        else:
            run(True)
        torch.cuda.synchronize()
    finally:
        _lib.call("adap_conv2d_set_clock_probe", 0)
        _lib.call("adap_conv2d_debug_force", 0, 0)
    c = clk.cpu().numpy().astype(np.float64) * 0.01          # us
    rows = []
    for i in range(2 * n):
        live = c[i][c[i, :, 0] > 0]
        if not len(live):
            continue
        rows.append((live[:, 0].min(), live[:, 0].max(), np.median(live[:, 1] - live[:, 0]), np.median(live[:, 2] - live[:, 1]),
                     np.median(live[:, 3] - live[:, 2]), live[:, 3].max(), len(live)))
    rows = rows[8:]                                        # steady state
    gaps = [rows[i + 1][0] - rows[i][5] for i in range(len(rows) - 1)]
    spans = [r[5] - r[0] for r in rows]
    period = (rows[-1][0] - rows[0][0]) / (len(rows) - 1)
    print(f"M={M} N={N} K={K} kind={kind} bn={bn} ks={ks} graph={graph}: {rows[0][6]} wgs; period {period:6.2f} us = "
          f"span {np.mean(spans):6.2f} (start spread {np.mean([r[1] - r[0] for r in rows]):5.2f}, first tile {np.mean([r[2] for r in rows]):5.2f}, "
          f"K loop {np.mean([r[3] for r in rows]):5.2f}, epilogue+drain {np.mean([r[4] for r in rows]):5.2f}) + gap {np.mean(gaps):5.2f}", flush=True)


for graph in (True, False):
    timeline(16384, 320, 320, 3, 160, 1, graph=graph)
    timeline(16384, 320, 320, 3, 64, 1, graph=graph)
    timeline(4096, 640, 640, 3, 64, 1, graph=graph)
    timeline(4096, 640, 640, 3, 128, 1, graph=graph)
    timeline(1024, 1280, 1280, 3, 64, 1, graph=graph)
    timeline(1024, 1280, 1280, 3, 160, 1, graph=graph)
    timeline(256, 1280, 1280, 3, 64, 1, graph=graph)
    timeline(16384, 2560, 320, 2, 160, 1, graph=graph)
