#!/usr/bin/env python
"""Host-side timeline of consecutive fused batch_filter calls (development tool): when does the host enter / leave each
call relative to the GPU's progress - i.e. does the host run ahead of the device or does something block it."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _env  # noqa: E402  (tools/_env.py: PF_AMD_LIB / PF_* of this process -> the package's explicit switches)

_env.setup()
from tools.kbench import make  # noqa: E402
from pyfilter_amd import _lib as L  # noqa: E402

f, _ = make("sine", "apf", "lgo", 1 << 20, 1)
T = 250
y = (0.3 * torch.randn(T)).cumsum(0).cuda()
for _ in range(3):
    f.batch_filter(y, bar=False)
torch.cuda.synchronize()
lib = L.load()
orig = lib.pf_filter_graph_launch
marks = []


def timed_launch(*a):
    t0 = time.perf_counter()
    rc = orig(*a)
    marks.append((t0, time.perf_counter()))
    return rc


lib.pf_filter_graph_launch = timed_launch
t_start = time.perf_counter()
rows = []
for i in range(6):
    t0 = time.perf_counter()
    f.batch_filter(y, bar=False)
    rows.append((t0, time.perf_counter()))
torch.cuda.synchronize()
t_end = time.perf_counter()
for (a, b), (la, lb) in zip(rows, marks):
    print(f"call: enter {1e6 * (a - t_start):8.0f} us  graph launch at {1e6 * (la - t_start):8.0f} (took {1e6 * (lb - la):6.0f})  "
          f"exit {1e6 * (b - t_start):8.0f}  host time in call {1e6 * (b - a):6.0f}")
print(f"all done (synced) at {1e6 * (t_end - t_start):8.0f} us -> {1e6 * (t_end - t_start) / 6:7.0f} us per call, {1e6 * (t_end - t_start) / 6 / T:6.2f} us per step")
