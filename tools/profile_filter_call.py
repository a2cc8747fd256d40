#!/usr/bin/env python
"""Host-side profile of one online ``filter()`` move (development tool): where the ~85 us of Python / launch time go."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.kbench import make  # noqa: E402

f, _ = make("sine", "apf", "lgo", 4096, 1)
state = f.initialize()
y = torch.tensor(0.1, device="cuda")
for _ in range(20):
    state = f.filter(y, state)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    state = f.filter(y, state)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
