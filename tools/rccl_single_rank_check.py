#!/usr/bin/env python3
"""RCCL sanity on a one-GPU box: a 1-rank "nccl" process group, an async 256 MB all-reduce, a barrier and a MAX
reduction -- the API path bench.py uses at N > 1 (multi-rank semantics are covered by the gloo tests)."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533"); os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY","0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
x = torch.ones(64<<20, device="cuda")
w = dist.all_reduce(x, op=dist.ReduceOp.SUM, async_op=True); w.wait()
dist.barrier(); torch.cuda.synchronize()
t = torch.tensor([1.5], device="cuda", dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
print("rccl 1-rank ok", float(x.sum()), float(t))
dist.destroy_process_group()
