#!/usr/bin/env python3
"""VAE encoder (bs=4, 512x512) alone, for rocprofv3 --stats: which kernels make up the first stage's ~10 ms."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
ld, hook = bench.build_model(dev)
batch = bench.synthetic_batch(4, dev, 1)
for _ in range(2):
    ld.get_input(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    ld.get_input(batch)
torch.cuda.synchronize()
print(f"VAE encode bs=4: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")
