#!/usr/bin/env python
"""cProfile of the reference-style SMC2.step() loop (one observation per call, host ESS test per observation):
python tools/smc2_step_profile.py [n_theta] [n_state] [T]"""
import cProfile
import math
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _env  # noqa: E402

_env.setup()


def main():
    from torch.distributions import Exponential, LogNormal, Normal

    from pyfilter_amd import timeseries as ts
    from pyfilter_amd.filters.particle import APF, proposals
    from pyfilter_amd.inference import SMC2
    from pyfilter_amd.timeseries import models

    n_theta = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    n_state = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    t_len = int(sys.argv[3]) if len(sys.argv) > 3 else 500
    device, dtype = torch.device("cuda"), torch.float32
    g = torch.Generator().manual_seed(123)
    x, ys = 0.0, []
    for _ in range(t_len):
        x = x * math.exp(-0.025) + 0.05 * math.sqrt((1 - math.exp(-0.05)) / 0.05) * torch.randn((), generator=g).item()
        ys.append(x + 0.05 * torch.randn((), generator=g).item())
    y = torch.tensor(ys, dtype=dtype, device=device)
    priors = {"kappa": Exponential(10.0), "gamma": Normal(0.0, 1.0), "sigma": LogNormal(-2.0, 1.0)}
    obs_a, obs_s = torch.tensor(1.0, dtype=dtype, device=device), torch.tensor(0.05, dtype=dtype, device=device)

    def build(theta):
        return ts.LinearStateSpaceModel(models.OrnsteinUhlenbeck(theta["kappa"], theta["gamma"], theta["sigma"], dt=1.0), (obs_a, obs_s))

    def run(seed, prof=None):
        filt = APF(build, n_state, proposal=proposals.LinearGaussianObservations(), seed=2024 + seed)
        alg = SMC2(filt, n_theta, priors, threshold=0.2, device=device, dtype=dtype, seed=seed)
        state = alg.initialize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if prof:
            prof.enable()
        for yt in y:
            state = alg.step(yt, state)
        torch.cuda.synchronize()
        if prof:
            prof.disable()
        return time.perf_counter() - t0, len(alg._kernel.acceptance_history)

    run(0)
    for s in (1, 2):
        dt, mv = run(s)
        print(f"step() loop at {n_theta} x {n_state}, T = {t_len}: {1e3 * dt:.1f} ms ({1e6 * dt / t_len:.1f} us per observation), PMMH moves {mv}")
    pr = cProfile.Profile()
    run(3, pr)
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(45)
    st.sort_stats("tottime").print_stats(30)


if __name__ == "__main__":
    main()
