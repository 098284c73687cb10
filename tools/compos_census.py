#!/usr/bin/env python3
"""launching aten ops of one compositional micro-batch (fresh + reuse averaged), grouped by the innermost repo source line."""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode

import bench
from adaprompt_amd import _lib

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
_lib.load()
NOLAUNCH = ("view", "reshape", "permute", "transpose", "expand", "slice", "select", "unsqueeze", "squeeze", "detach", "alias", "as_strided",
            "t.default", "empty", "_unsafe_view", "unbind", "split", "chunk", "size", "stride", "is_", "_local_scalar", "item", "numel",
            "record_stream", "lift_fresh", "_reshape_alias", "narrow", "unfold", "contiguous", "requires_grad", "dim", "sym_")
counts = collections.Counter()
on = [False]


class Census(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if on[0] and not any(s in name for s in NOLAUNCH):
            site = "?"
            for fr in reversed(traceback.extract_stack()):
                if ROOT in fr.filename and "tools/" not in fr.filename and "bench.py" not in fr.filename:
                    site = f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno} {fr.name}"
                    break
            counts[(name.replace("aten.", ""), site)] += 1
        return func(*args, **(kwargs or {}))


orig = bench.compos_leg


def patched_timer():
    pass


gen = torch.Generator(device=dev).manual_seed(99)
import time
_pc = time.perf_counter
state = {"n": 0}


def pc():
    state["n"] += 1
    on[0] = state["n"] == 1          # between the leg's two perf_counter() calls = the timed micro-batches
    return _pc()


bench.time.perf_counter = pc
with Census():
    res = bench.compos_leg(dev, gen, micro_batches=2)
on[0] = False
tot = sum(counts.values())
print("launching aten ops per micro-batch:", tot / 2)
by_site = collections.Counter()
for (op, site), n in counts.items():
    by_site[site] += n
for site, n in by_site.most_common(40):
    print(f"{n / 2:8.1f}  {site}")
