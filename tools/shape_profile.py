#!/usr/bin/env python3
"""Per-shape timing of the contraction / attention launches of one training step (tuning aid, GPU only)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from adaprompt_amd import ops
from adaprompt_amd.parallel import GradReducer

dev = torch.device("cuda:0")
ld, hook = bench.build_model(dev)
params = list(hook.parameters())
red = GradReducer(params)
batch = bench.synthetic_batch(4, dev, 1)


def step():
    t = torch.randint(0, 1000, (4,), device=dev)
    loss, grad, out, aux = ld.shared_step(batch, t=t, noise=torch.randn(4, 4, 64, 64, device=dev),
                                          post_noise=torch.randn(4, 4, 64, 64, device=dev))
    out.backward(grad)


for _ in range(2):
    step()
ops.TIMER = ops.KernelTimer()
for _ in range(2):
    step()
rows = ops.TIMER.by_tag()
ops.TIMER = None
tot = sum(v["ms"] for v in rows.values()) / 2
print(f"timed families total {tot:.2f} ms/step")
for (fam, tag), v in sorted(rows.items(), key=lambda kv: -kv[1]["ms"]):
    unit = 1e9 if fam == "groupnorm_fwd" else 1e12
    print(f"{fam:14s} {str(tag):44s} n/step {v['launches'] // 2:3d}  ms/step {v['ms'] / 2:7.3f}  "
          f"avg us {1e3 * v['ms'] / v['launches']:8.1f}  {v['work'] / (v['ms'] * 1e-3) / unit:8.1f} {'GB/s' if unit == 1e9 else 'TF/s'}")
