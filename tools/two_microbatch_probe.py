#!/usr/bin/env python3
"""What would issuing the two micro-batches of one accumulation window on TWO streams buy?  (Both read the same weights --
the optimiser steps every 2nd micro-batch, ddpm.py:606-633 -- so they are independent.)  Timing only: the hook gradients of the
two streams race into one buffer here; a real version would give each stream its own gradient buffer.
  A: the bench's loop (one stream + VAE prefetch stream)
  B: even micro-batches on stream 0, odd ones on stream 1 (GroupNorm two-pass on stream 1: the single-launch kernel's
     in-launch exchange belongs to one stream per device), joined before the optimiser step."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from adaprompt_amd.ldm.prodigy import Prodigy

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
ld, hook = bench.build_model(dev)
params = list(hook.parameters())
opt = Prodigy(params, lr=1.0, betas=(0.9, 0.999), d_coef=2.0, use_bias_correction=True, weight_decay=0.0)
B = 4
batches = [bench.synthetic_batch(B, dev, 1234 + i) for i in range(2)]
gen = torch.Generator(device=dev).manual_seed(99)
pf = ld.make_prefetcher()


def micro(i, x_start):
    t = torch.randint(0, 1000, (B,), device=dev, generator=gen)
    noise = torch.randn(B, 4, 64, 64, device=dev, generator=gen)
    loss, grad, out, aux = ld.shared_step(batches[i % 2], t=t, noise=noise, x_start=x_start, anneal_t=True)
    ld.manual_backward(out, grad, aux)
    return loss


def submit(i):
    pf.submit(batches[i % 2], torch.randn(B, 4, 64, 64, device=dev, generator=gen))


def run_single(n):
    submit(0)
    for i in range(n):
        x = pf.get()
        submit(i + 1)
        micro(i, x)
        if i % 2 == 1:
            opt.step(clip_norm=0.5)
            opt.zero_grad(set_to_none=False)
    pf.get()


s1 = torch.cuda.Stream()


def run_dual(n):
    main = torch.cuda.current_stream()
    submit(0)
    for i in range(0, n, 2):
        x0 = pf.get()
        submit(i + 1)
        s1.wait_stream(main)
        with torch.cuda.stream(s1):            # the odd micro-batch first: its host issue overlaps nothing yet
            x1 = pf.get()
        submit(i + 2)
        micro(i, x0)
        with torch.cuda.stream(s1):
            micro(i + 1, x1)
        main.wait_stream(s1)
        opt.step(clip_norm=0.5)
        opt.zero_grad(set_to_none=False)
    pf.get()


def timed(fn, n):
    fn(4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(n)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


for rep in range(2):
    a = timed(run_single, 12)
    b = timed(run_dual, 12)
    print(f"rep {rep}: one stream {a:.2f} ms/micro-batch   two streams {b:.2f} ms/micro-batch", flush=True)
