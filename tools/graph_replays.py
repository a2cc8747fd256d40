#!/usr/bin/env python
"""Replays one captured hipGraph of a short fused run many hundred times and checks every log-likelihood (development /
regression tool).  Background: with the per-column records of a fresh run cleared by ``hipMemsetAsync`` - a memset node in
the captured graph - every run from the ~196th replay of the same executable graph on returned NaN log-likelihoods (the
"poisoned" flags were no longer cleared) while particles, weights and moments stayed correct; direct launches with the
same arguments and a re-captured graph were fine.  The fill is a kernel now.
Usage: python tools/graph_replays.py [replays]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _env  # noqa: E402  (tools/_env.py: PF_AMD_LIB / PF_* of this process -> the package's explicit switches)

_env.setup()


def first_bad_replay(steps=2, b=256, n=2048, replays=600, seed=7):
    from torch.distributions import Exponential, LogNormal, Normal

    from pyfilter_amd import timeseries as ts
    from pyfilter_amd.filters.particle import APF, proposals
    from pyfilter_amd.inference import ThetaParticles
    from pyfilter_amd.timeseries import models

    device, dtype = torch.device("cuda"), torch.float32
    g = torch.Generator().manual_seed(123)
    x, ys = 0.0, []
    for _ in range(steps * replays):
        x = x * math.exp(-0.025) + 0.05 * math.sqrt((1 - math.exp(-0.05)) / 0.05) * torch.randn((), generator=g).item()
        ys.append(x + 0.05 * torch.randn((), generator=g).item())
    y = torch.tensor(ys, dtype=dtype, device=device)
    theta = ThetaParticles({"kappa": Exponential(10.0), "gamma": Normal(0.0, 1.0), "sigma": LogNormal(-2.0, 1.0)}, b, device, dtype)
    theta.initialize_parameters(torch.Generator().manual_seed(3))

    def build(th):
        t = lambda v: torch.tensor(v, dtype=dtype, device=device)  # noqa: E731
        return ts.LinearStateSpaceModel(models.OrnsteinUhlenbeck(th["kappa"], th["gamma"], th["sigma"], dt=1.0), (t(1.0), t(0.05)))

    filt = APF(build, n, proposal=proposals.LinearGaussianObservations(), seed=seed)
    filt.set_batch_shape(torch.Size([b]))
    filt.initialize_model(theta)
    s = filt.initialize()
    flags = torch.ones(steps * replays, dtype=torch.uint8)
    bad = torch.zeros((), device=device)
    plans = set()
    for k in range(replays):
        t = steps * k
        res, ll, _ = filt.filter_block(y[t:t + steps], s, observed=flags[t:t + steps])
        plans.add(id(filt._last_run["plan"]))
        bad = bad + (~torch.isfinite(ll)).any() * (bad == 0) * (k + 1)  # first failing replay (1-based), no sync per replay
        s = res.latest_state
    assert len(plans) == 1, "the runs were meant to replay one cached plan"
    first = int(bad.item())
    return None if first == 0 else first - 1


if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    for cfg in ((1, 256, 2048), (2, 256, 2048), (2, 1024, 8192), (4, 64, 2048)):
        print("steps, B, N =", cfg, "-> first replay with a non-finite log-likelihood:", first_bad_replay(*cfg, replays=reps), flush=True)
