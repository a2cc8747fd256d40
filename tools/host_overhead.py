#!/usr/bin/env python
"""Host-side cost of one fused ``batch_filter`` call (development tool): wall time per call as a function of T - the
intercept is what a call costs beyond its T step kernels (staging copies, graph launch, result hand-over)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.kbench import make  # noqa: E402

for cfg in (("sine", "apf", "lgo", 1 << 20, 1), ("sine", "apf", "lgo", 8192, 128)):
    f, _ = make(*cfg)
    out = []
    for T in (1, 50, 250):
        y = (0.3 * torch.randn(T)).cumsum(0).cuda()
        for _ in range(3):
            f.batch_filter(y, bar=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            f.batch_filter(y, bar=False)
        torch.cuda.synchronize()
        out.append((T, 1e6 * (time.perf_counter() - t0) / reps))
    print(cfg, " ".join(f"T={t}: {us:.0f} us" for t, us in out), flush=True)
import cProfile, pstats
y = (0.3 * torch.randn(1)).cumsum(0).cuda()
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    f.batch_filter(y, bar=False)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
