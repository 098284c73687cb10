#!/usr/bin/env python3
"""What would two half-batch streams buy?  hipGraph replay of the UNet fwd + losses + bwd (no VAE, no optimiser) for the whole
micro-batch (B = 4, one graph on one stream) against two B = 2 graphs replayed CONCURRENTLY on two streams."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
ld, hook = bench.build_model(dev)
gen = torch.Generator(device=dev).manual_seed(1)


def make(B, seed):
    batch = bench.synthetic_batch(B, dev, seed)
    x0 = torch.randn(B, 4, 64, 64, device=dev, generator=gen)
    t = torch.randint(0, 1000, (B,), device=dev, generator=gen)
    noise = torch.randn(B, 4, 64, 64, device=dev, generator=gen)

    def one():
        loss, grad, out, aux = ld.shared_step(batch, t=t, noise=noise, x_start=x0)
        ld.manual_backward(out, grad, aux)
        return loss
    return one


def capture(fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    torch.cuda.synchronize()
    return g


def timed(run, n=10):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        run()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


def say(*a):
    print(*a, flush=True)


f4 = make(4, 11)
eager4 = timed(f4)
say("eager B=4", eager4)
g4 = capture(f4)
say("captured B=4")
t4 = timed(g4.replay)
say("graph B=4", t4)
fa, fb = make(2, 12), make(2, 13)
ga = capture(fa)
say("captured a")
gb = capture(fb)
say("captured b")
ta = timed(ga.replay)
say("graph B=2", ta)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def both():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        ga.replay()
    with torch.cuda.stream(s2):
        gb.replay()
    cur.wait_stream(s1)
    cur.wait_stream(s2)


t2s0 = None
say('two streams ...')
t22 = timed(both)
say('two streams', t22)


def serial():
    ga.replay()
    gb.replay()


t2s = timed(serial)
print(f"UNet fwd + losses + bwd: B=4 eager {eager4:.2f} ms | B=4 graph {t4:.2f} ms | one B=2 graph {ta:.2f} ms | "
      f"two B=2 graphs back to back {t2s:.2f} ms | two B=2 graphs on two streams {t22:.2f} ms")
