#!/usr/bin/env python
"""Where an observation of the fast SMC2.step() loop (_OnlineRun) spends its wall time: host work before the C call, the C call
(pf_filter_observe), the wait for the host slot, everything outside observe().  python tools/smc2_fast_segments.py [n_theta] [n_state] [T]"""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _env  # noqa: E402

_env.setup()


def main():
    from torch.distributions import Exponential, LogNormal, Normal

    from pyfilter_amd import ops
    from pyfilter_amd import timeseries as ts
    from pyfilter_amd.filters.particle import APF, proposals
    from pyfilter_amd.filters.particle import base as pbase
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

    acc = {"observe: before the wait": 0.0, "observe: the wait (slot)": 0.0, "update (rejuvenations)": 0.0}
    real_wait = ops.HostSlot.wait
    real_observe = pbase._OnlineRun.observe
    mark = [0.0]

    def wait(self):
        t1 = time.perf_counter()
        acc["observe: before the wait"] += t1 - mark[0]
        r = real_wait(self)
        acc["observe: the wait (slot)"] += time.perf_counter() - t1
        return r

    def observe(self, *a, **k):
        mark[0] = time.perf_counter()
        return real_observe(self, *a, **k)

    ops.HostSlot.wait = wait
    pbase._OnlineRun.observe = observe
    for seed in range(3):
        for k_ in acc:
            acc[k_] = 0.0
        alg = SMC2(APF(build, n_state, proposal=proposals.LinearGaussianObservations(), seed=2024 + seed), n_theta, priors, threshold=0.2,
                   device=device, dtype=dtype, seed=seed)
        state = alg.initialize()
        upd = alg._kernel.update

        def timed_update(*a, **k):
            t1 = time.perf_counter()
            try:
                return upd(*a, **k)
            finally:
                acc["update (rejuvenations)"] += time.perf_counter() - t1
        alg._kernel.update = timed_update
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for yt in y:
            state = alg.step(yt, state)
        torch.cuda.synchronize()
        total = time.perf_counter() - t0
    rest = total - sum(acc.values())
    print(f"fast step() loop at {n_theta} x {n_state}, T = {t_len}: {1e3 * total:.1f} ms; per observation (us):")
    for k_, v in acc.items():
        print(f"   {k_:36s} {1e6 * v / t_len:7.2f}" + ("   (ms in total: %.2f)" % (1e3 * v) if "rejuv" in k_ else ""))
    print(f"   {'outside observe() (step, _step, ...)':36s} {1e6 * rest / t_len:7.2f}")


if __name__ == "__main__":
    main()
