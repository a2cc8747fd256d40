#!/usr/bin/env python
"""Where an SMC^2 fit spends its time (development tool): per-observation wall time of the online move and of the
rejuvenations, BASELINE configs[4] (1024 theta x 8192 particles, T = 500).  Every step ends with the algorithm's one
device -> host copy (the ESS decision), so ``perf_counter`` per step is meaningful.
Usage: python tools/smc2_timeline.py [T]"""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _env  # noqa: E402  (tools/_env.py: PF_AMD_LIB / PF_* of this process -> the package's explicit switches)

_env.setup()


def main():
    from torch.distributions import Exponential, LogNormal, Normal

    from pyfilter_amd import timeseries as ts
    from pyfilter_amd.filters.particle import APF, proposals
    from pyfilter_amd.inference import SMC2
    from pyfilter_amd.timeseries import models

    t_len = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    device, dtype = torch.device("cuda"), torch.float32
    g = torch.Generator().manual_seed(123)
    x, ys = 0.0, []
    for _ in range(t_len):
        x = x * math.exp(-0.025) + 0.05 * math.sqrt((1 - math.exp(-0.05)) / 0.05) * torch.randn((), generator=g).item()
        ys.append(x + 0.05 * torch.randn((), generator=g).item())
    y = torch.tensor(ys, dtype=dtype, device=device)
    priors = {"kappa": Exponential(10.0), "gamma": Normal(0.0, 1.0), "sigma": LogNormal(-2.0, 1.0)}

    # the observation constants live on the device ONCE: created inside the builder they would be two pageable host -> device
    # copies - two waits for the device - per model build, and PMMH rebuilds the model at every move
    obs_a, obs_s = torch.tensor(1.0, dtype=dtype, device=device), torch.tensor(0.05, dtype=dtype, device=device)

    def build(theta):
        return ts.LinearStateSpaceModel(models.OrnsteinUhlenbeck(theta["kappa"], theta["gamma"], theta["sigma"], dt=1.0), (obs_a, obs_s))

    for rep in range(3):
        filt = APF(build, 8192, proposal=proposals.LinearGaussianObservations(), seed=2024 + rep)
        alg = SMC2(filt, 1024, priors, threshold=0.2, device=device, dtype=dtype, seed=rep)
        state = alg.initialize()
        torch.cuda.synchronize()
        online, rejuv = [], []
        t_all = time.perf_counter()
        for t, yt in enumerate(y):
            n0 = len(alg._kernel.acceptance_history)
            t0 = time.perf_counter()
            state = alg.step(yt, state)
            dt = time.perf_counter() - t0
            (rejuv if len(alg._kernel.acceptance_history) > n0 else online).append((t, dt))
        torch.cuda.synchronize()
        total = time.perf_counter() - t_all
        on = sorted(d for _, d in online)
        print(f"rep {rep}: total {1e3 * total:7.1f} ms | online moves {len(on)}: sum {1e3 * sum(on):7.1f} ms, median {1e6 * on[len(on) // 2]:6.1f} us, "
              f"p90 {1e6 * on[int(0.9 * len(on))]:6.1f} us | rejuvenations {len(rejuv)}: " +
              ", ".join(f"t={t} {1e3 * d:.1f} ms" for t, d in rejuv), flush=True)

    # fit(): the filters run `block` observations ahead of the rejuvenation test (one host decision point per block)
    for block in (1, 4, 8, 16, 32, 64):
        best = None
        for rep in range(3):
            filt = APF(build, 8192, proposal=proposals.LinearGaussianObservations(), seed=2024 + rep)
            alg = SMC2(filt, 1024, priors, threshold=0.2, device=device, dtype=dtype, seed=rep)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            state = alg.fit(y, block=block)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        print(f"fit(block={block:2d}): best of 3 {1e3 * best:7.1f} ms  ({1024 * 8192 * t_len / best:.3e} particle-steps/s), "
              f"rejuvenations {len(alg._kernel.acceptance_history)}, posterior mean {[round(v, 4) for v in alg.posterior_mean(state).tolist()]}", flush=True)


if __name__ == "__main__":
    main()
