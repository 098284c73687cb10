#!/usr/bin/env python3
"""Is the training step host-bound?  Issues N full micro-batches (VAE prefetch stream, UNet fwd + aux losses + bwd, Prodigy every
2nd) without any synchronisation and reports how long the HOST needed to issue them against how long the GPU needed to
finish them.  issue ~= total: the host is the bottleneck (the GPU drains right behind it)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
from adaprompt_amd import _lib
from adaprompt_amd.ldm.prodigy import Prodigy
from adaprompt_amd.ldm.util import prodigy_linear_schedule
from adaprompt_amd.parallel import GradReducer

_lib.load()
ld, hook = bench.build_model(dev)
params = list(hook.parameters())
opt = Prodigy(params, lr=1.0, betas=(0.9, 0.999), d_coef=2.0, use_bias_correction=True, weight_decay=0.0)
red = GradReducer(params, flat=opt.grad_buffer)
sched = prodigy_linear_schedule(opt, max_steps=60000, warm_up_steps=500, scheduler_cycles=1)
B = 4
batches = [bench.synthetic_batch(B, dev, 1234 + i) for i in range(2)]
gen = torch.Generator(device=dev).manual_seed(99)
pf = ld.make_prefetcher()
pf.submit(batches[0], torch.randn(B, 4, 64, 64, device=dev, generator=gen))
marks = {}


def step(i, stamp=False):
    t0 = time.perf_counter()
    t = torch.randint(0, 1000, (B,), device=dev, generator=gen)
    noise = torch.randn(B, 4, 64, 64, device=dev, generator=gen)
    x_start = pf.get()
    loss, grad, out, aux = ld.shared_step(batches[i % 2], t=t, noise=noise, x_start=x_start, anneal_t=True)
    t1 = time.perf_counter()
    pf.submit(batches[(i + 1) % 2], torch.randn(B, 4, 64, 64, device=dev, generator=gen))
    t2 = time.perf_counter()
    red.wait()
    ld.manual_backward(out, grad, aux)
    t3 = time.perf_counter()
    red.reduce()
    ld.batch_idx += 1
    if ld.batch_idx % 2 == 0:
        opt.step(clip_norm=ld.grad_clip)
        red.zero()
        sched.step()
    t4 = time.perf_counter()
    if stamp:
        for k, v in (("fwd+losses", t1 - t0), ("vae submit", t2 - t1), ("backward", t3 - t2), ("optim", t4 - t3)):
            marks[k] = marks.get(k, 0.0) + v


for i in range(4):
    step(i)
torch.cuda.synchronize()
N = 10
t0 = time.perf_counter()
for i in range(N):
    step(i, True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{N} steps: host issue {1e3 * (t1 - t0) / N:.2f} ms/step, issue + drain {1e3 * (t2 - t0) / N:.2f} ms/step "
      f"(drain after the last issue: {1e3 * (t2 - t1):.2f} ms)")
print("host ms/step by phase:", {k: round(1e3 * v / N, 2) for k, v in marks.items()})

# the host's own work per step: with an empty launch queue (a synchronisation first) and only a step or two issued, the host
# never waits for the GPU; over ten steps the queue fills (it holds ~55 ms of this workload) and "issue" time becomes
# (GPU time - queue depth) / steps, whatever the host could do
for n in (1, 2):
    ts = []
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            step(i)
        ts.append((time.perf_counter() - t0) / n)
    torch.cuda.synchronize()
    print(f"host work per step, {n} step(s) issued into an empty queue: {1e3 * sorted(ts)[len(ts) // 2]:.2f} ms (median of 5)")

if os.environ.get("HOST_PROFILE"):
    import cProfile
    import pstats
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    for i in range(4):
        step(i)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(45)
    st.sort_stats("cumtime").print_stats(60)
