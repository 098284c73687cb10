#!/usr/bin/env python3
"""When do the two lanes of a window start and finish on the GPU?  Events at the window's start (lane 0), after each lane's forward
and after each lane's backward; prints their times relative to the window start, averaged over the timed windows."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from adaprompt_amd.ldm.models.diffusion.ddpm import MicroBatchLanes
from adaprompt_amd.ldm.prodigy import Prodigy

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
ld, hook = bench.build_model(dev)
params = list(hook.parameters())
opt = Prodigy(params, lr=1.0, betas=(0.9, 0.999), d_coef=2.0, use_bias_correction=True, weight_decay=0.0)
B = 4
batches = [bench.synthetic_batch(B, dev, 1234 + i) for i in range(2)]
gen = torch.Generator(device=dev).manual_seed(99)
lanes = MicroBatchLanes(params, n=2)
pf = ld.make_prefetcher()


def submit(i):
    pf.submit(batches[i % 2], torch.randn(B, 4, 64, 64, device=dev, generator=gen))


for i in range(4):
    submit(i)
rows = []
hrows = []


def window(i, record):
    ev = {}

    hs = {}

    def mark(name):
        hs[name] = time.perf_counter()                # host clock at the moment the mark is ISSUED
        if record:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            ev[name] = e

    def draws(k):
        if k == 0:
            mark("start")
        mark(f"f{k}_begin")
        return dict(t=torch.randint(0, 1000, (B,), device=dev, generator=gen), noise=torch.randn(B, 4, 64, 64, device=dev, generator=gen),
                    x_start=pf.get(), anneal_t=True)
    th0 = time.perf_counter()
    ld.training_window([batches[(i + k) % 2] for k in range(2)], opt, None, None, lanes, step_kwargs=draws,
                       after_forward=lambda k: mark(f"f{k}_end"), after_backward=lambda k: (mark(f"b{k}_end"), submit(i + 4 + k)))
    mark("opt_end")
    host = 1e3 * (time.perf_counter() - th0)
    if record:
        rows.append((ev, host))
        hrows.append({n: 1e3 * (t - th0) for n, t in hs.items()})


for w in range(3):
    window(2 * w, False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for w in range(8):
    window(2 * w, True)
torch.cuda.synchronize()
print(f"{1e3 * (time.perf_counter() - t0) / 16:.2f} ms per micro-batch")
names = ["f0_begin", "f1_begin", "f0_end", "f1_end", "b0_end", "b1_end", "opt_end"]
acc = {n: 0.0 for n in names}
for ev, _ in rows:
    for n in names:
        acc[n] += ev["start"].elapsed_time(ev[n])
print("GPU time since the window's start (ms, mean of 8 windows):", {n: round(acc[n] / len(rows), 2) for n in names})
print("host issue time per window (ms):", round(sum(h for _, h in rows) / len(rows), 2))
print("host clock when each mark was issued (ms since the window call, mean):",
      {n: round(sum(h[n] for h in hrows) / len(hrows), 2) for n in ["start"] + names})
