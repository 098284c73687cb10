#!/usr/bin/env python3
"""timing of the stand-in hook's mixture contraction in a few formulations (fwd + bwd), GPU only"""
import torch
dev = torch.device("cuda:0")
G, P, B = 157, 16 * 77 * 768, 4
bases = (torch.randn(G, P, device=dev) * 0.05).requires_grad_(True)
a0 = torch.softmax(torch.randn(B, G, device=dev), -1)
gout = torch.randn(B, P, device=dev)
basesT = bases.detach().t().contiguous().requires_grad_(True)


def t(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def v1():
    a = a0.clone().requires_grad_(True)
    (a @ bases).backward(gout)


def v2():
    a = a0.clone().requires_grad_(True)
    (bases.t() @ a.t()).t().backward(gout)


def v3():
    a = a0.clone().requires_grad_(True)
    torch.nn.functional.linear(a, basesT).backward(gout)


def v4():          # per-instance axpy accumulation: element-wise, HBM-bound
    a = a0.clone().requires_grad_(True)
    (a.unsqueeze(-1) * bases.unsqueeze(0)).sum(1).backward(gout)


for name, fn in (("a @ bases", v1), ("(bases^T @ a^T)^T", v2), ("linear(a, bases^T stored)", v3), ("broadcast mul + sum", v4)):
    bases.grad = None
    basesT.grad = None
    print(f"{name:28s} {t(fn):8.1f} us fwd+bwd")
