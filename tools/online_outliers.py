#!/usr/bin/env python
"""Per-move wall time of an online filter() loop (synchronised per move): outliers and non-finite log-likelihoods.
python tools/online_outliers.py model filter proposal N B [moves]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kbench import make  # noqa: E402

cfg = (sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]))
moves = int(sys.argv[6]) if len(sys.argv) > 6 else 300
f, _ = make(*cfg)
state = f.initialize()
y = torch.tensor(0.1, device="cuda")
ts, bad = [], 0
from pyfilter_amd import ops
for i in range(moves):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    state = f.filter(y, state)
    torch.cuda.synchronize()
    ts.append(1e6 * (time.perf_counter() - t0))
    if not bool(torch.isfinite(state.get_loglikelihood()).all()):
        bad += 1
srt = sorted(ts)
print(cfg, "median %.1f us, p90 %.1f, max %.1f; moves over 300 us:" % (srt[len(srt) // 2], srt[int(0.9 * len(srt))], srt[-1]),
      [(i, round(t)) for i, t in enumerate(ts) if t > 300][:20], "non-finite ll:", bad, "trace", ops.debug_launch_trace(1)[-1])
