#!/usr/bin/env python3
"""The first stage alone: ``get_input`` (VAE encode of a bs-4 512 x 512 batch + posterior sample) ITERS times on one stream, timed
with events; under ``rocprofv3 --kernel-trace --stats`` the per-kernel table of the encoder by itself (which of the step's
chip-filling work is the encoder's, and where inside it)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

dev = torch.device("cuda:0")
ITERS = int(os.environ.get("ITERS", "20"))
ld, hook = bench.build_model(dev)
batch = bench.synthetic_batch(int(os.environ.get("BATCH", "4")), dev, 1234)
pn = torch.randn(batch["image"].shape[0], 4, 64, 64, device=dev)
for _ in range(5):
    ld.get_input(batch, pn)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(ITERS):
    ld.get_input(batch, pn)
e1.record()
torch.cuda.synchronize()
print(f"VAE encode + sample, bs {batch['image'].shape[0]}: {e0.elapsed_time(e1) / ITERS:.3f} ms per call", flush=True)
