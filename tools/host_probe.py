#!/usr/bin/env python3
"""How long does the HOST take to issue one micro-batch (all launches asynchronous) vs. the GPU to run it?
If issue time approaches GPU time the step is launch-bound and hipGraph replay (bench.py --graph) pays."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
ld, hook = bench.build_model(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
batch = bench.synthetic_batch(B, dev, 1)
gen = torch.Generator(device=dev).manual_seed(1)
x0 = torch.randn(B, 4, 64, 64, device=dev)


def one():
    t = torch.randint(0, 1000, (B,), device=dev, generator=gen)
    noise = torch.randn(B, 4, 64, 64, device=dev, generator=gen)
    loss, grad, out, aux = ld.shared_step(batch, t=t, noise=noise, x_start=x0)
    out.backward(grad)


for _ in range(3):
    one()
torch.cuda.synchronize()
issue, total = [], []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    one()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    issue.append(t1 - t0)
    total.append(t2 - t0)
print(f"B={B}: UNet fwd+bwd (no VAE): host issue {1e3 * min(issue):.1f} ms, issue+drain {1e3 * min(total):.1f} ms")

if os.environ.get("HOST_PROFILE"):
    import cProfile
    import pstats
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    for _ in range(3):
        one()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)
