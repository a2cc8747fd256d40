#!/usr/bin/env python
"""Latency of one online ``filter()`` move: fused single-step path vs the step-by-step route over the stand-alone kernels
(development tool).  Usage: python tools/step_latency.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kbench import make  # noqa: E402


def run(cfg, fused, reps=60):
    from pyfilter_amd.hints import HINTS

    HINTS.fused_step = bool(fused)
    f, _ = make(*cfg)
    state = f.initialize()
    y = torch.tensor(0.1, device="cuda")
    for _ in range(5):
        state = f.filter(y, state)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        state = f.filter(y, state)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for cfg in [("sine", "apf", "lgo", 1 << 20, 1), ("sine", "apf", "lgo", 65536, 64), ("sine", "sisr", "bootstrap", 8192, 1024),
            ("sine", "apf", "bootstrap", 4096, 1)]:
    a, b = run(cfg, True), run(cfg, False)
    print(f"{cfg}: fused single step {a:8.1f} us   step-by-step {b:8.1f} us   ({b / a:.2f}x)", flush=True)
