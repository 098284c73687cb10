#!/usr/bin/env python3
"""Which lines of this repo's host code issue torch-native device work in one training micro-batch (the launches that are not
this package's HIP kernels): aten ops of two steady-state steps, grouped by the innermost repo source line on their stack."""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import ProfilerActivity, profile

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


def step(i):
    t = torch.randint(0, 1000, (B,), device=dev, generator=gen)
    noise = torch.randn(B, 4, 64, 64, device=dev, generator=gen)
    x_start = pf.get()
    loss, grad, out, aux = ld.shared_step(batches[i % 2], t=t, noise=noise, x_start=x_start, anneal_t=True)
    pf.submit(batches[(i + 1) % 2], torch.randn(B, 4, 64, 64, device=dev, generator=gen))
    red.wait()
    ld.manual_backward(out, grad, aux)
    red.reduce()
    ld.batch_idx += 1
    if ld.batch_idx % 2 == 0:
        opt.step(clip_norm=ld.grad_clip)
        red.zero()
        sched.step()


import traceback

from torch.utils._python_dispatch import TorchDispatchMode

for i in range(4):
    step(i)
torch.cuda.synchronize()
NOLAUNCH = ("view", "reshape", "permute", "transpose", "expand", "slice", "select", "unsqueeze", "squeeze", "detach", "alias", "as_strided",
            "t.default", "empty", "_unsafe_view", "unbind", "split", "chunk", "size", "stride", "is_", "_local_scalar", "item", "numel",
            "record_stream", "lift_fresh", "_reshape_alias", "narrow", "unfold", "contiguous", "requires_grad", "dim", "sym_")
by_line = collections.Counter()


class Census(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(k in name for k in NOLAUNCH):
            where = "?"
            for fr in reversed(traceback.extract_stack(limit=40)):
                if ROOT in fr.filename and "op_census" not in fr.filename:
                    where = f"{fr.filename.replace(ROOT + '/', '')}:{fr.lineno} {fr.name}"
                    break
            shape = next((tuple(a.shape) for a in args if torch.is_tensor(a)), None)
            by_line[(where, name.replace("aten.", ""), shape if where == "?" else None)] += 1
        return func(*args, **(kwargs or {}))


with torch.autograd.set_multithreading_enabled(False), Census():
    for i in range(4, 6):
        step(i)
torch.cuda.synchronize()
print("launching aten ops per step:", sum(by_line.values()) / 2)
for (where, name, shape), n in by_line.most_common(120):
    print(f"{n / 2:7.1f}  {name:34s} {where} {shape if shape else ''}")
