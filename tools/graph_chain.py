#!/usr/bin/env python
"""What a chain of short dependent elementwise launches costs inside one captured hipGraph (the callable's share of a
graph_callable move): T x [sub, sin] and T x [sub, sin, add] on 2^20 floats, replayed."""
import time

import torch

dev = "cuda"
n, T = 1 << 20, 200
x = torch.randn(n, device=dev)
gm = torch.tensor(0.1, device=dev)


def chain(k):
    out = None
    for _ in range(T):
        f = torch.sin(x - gm)
        out = torch.add(x, f, alpha=0.1) if k == 3 else f
        if k == 1:
            pass
    return out


for k in (2, 3):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        chain(k)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            chain(k)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20 / T
    print(f"{k} elementwise launches per step: {1e6 * dt:.2f} us per step ({1e6 * dt / k:.2f} per launch)")
