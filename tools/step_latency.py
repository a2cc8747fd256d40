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
    slow = []
    for i in range(reps):
        t1 = time.perf_counter()
        state = f.filter(y, state)
        if os.environ.get("SL_TRACE") and time.perf_counter() - t1 > 3e-4:  # (host time of the call alone: no synchronisation added)
            slow.append((i, round(1e6 * (time.perf_counter() - t1))))
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    if os.environ.get("SL_TRACE"):
        print("   calls over 300 us of host time:", slow, " final synchronise: %.0f us" % (1e6 * (time.perf_counter() - t2)))
    return (time.perf_counter() - t0) / reps * 1e6


if os.environ.get("SL_NOGC"):
    import gc

    gc.disable()
if os.environ.get("SL_TIME_POOL"):
    from pyfilter_amd.filters.particle import base as _b

    _inner = _b._SingleStepPlan.zeroed_stats

    def _timed(self, batched):
        t = time.perf_counter()
        refill = self._pool is None or self._pool_next == self._STATS_POOL
        if refill:
            z0 = time.perf_counter()
            torch.zeros((64, 6, 1), device=self.device, dtype=self.dtype)
            z1 = time.perf_counter()
        r = _inner(self, batched)
        if refill:
            print("   pool refill: a torch.zeros of the pool's size alone %.0f us, zeroed_stats %.0f us" % (1e6 * (z1 - z0), 1e6 * (time.perf_counter() - z1)))
        return r
    _b._SingleStepPlan.zeroed_stats = _timed
for cfg in [("sine", "apf", "lgo", 1 << 20, 1), ("sine", "apf", "lgo", 65536, 64), ("sine", "sisr", "bootstrap", 8192, 1024),
            ("sine", "apf", "bootstrap", 4096, 1)]:
    a, b = run(cfg, True), run(cfg, False)
    print(f"{cfg}: fused single step {a:8.1f} us   step-by-step {b:8.1f} us   ({b / a:.2f}x)", flush=True)
