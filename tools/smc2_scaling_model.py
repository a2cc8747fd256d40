#!/usr/bin/env python
"""Expected strong-scaling efficiency of BASELINE configs[4] (SMC^2, 1 024 theta x 8 192 state particles, T = 500) over
2 / 4 / 8 GPUs, from ONE GPU (development tool; the driver measures the real curve when an 8-GPU node exists).

A rank of an N-GPU run does exactly what a single process does for 1024 / N theta-particles - the same fused blocks,
the same host decisions, the same rejuvenations over its own theta - plus the collectives.  So: time `SMC2.fit` on this
GPU with 1024, 512, 256, 128 theta-particles (everything a rank does except the collectives), add a priced estimate of
the collectives (one latency-bound all-gather of the block's (16, B / N) weight paths per block of 16 observations, and per
rejuvenation one all-gather of (B, P) theta values + weights and one all-to-all of the moved filters' states), and report
    efficiency(N) = T(1024) / (N * (T(1024 / N) + collectives(N))).
Usage: python tools/smc2_scaling_model.py [T]"""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _env  # noqa: E402  (tools/_env.py: PF_AMD_LIB / PF_* of this process -> the package's explicit switches)

_env.setup()

ALL_GATHER_US = 25.0     # small-message RCCL all-gather over xGMI, latency-bound (8 ranks, a few KiB)
XGMI_GBS = 153.0 * 0.8   # one point-to-point xGMI link at ~80 % of its 153 GB/s


def main():
    from torch.distributions import Exponential, LogNormal, Normal

    from pyfilter_amd import timeseries as ts
    from pyfilter_amd.filters.particle import APF, proposals
    from pyfilter_amd.inference import SMC2
    from pyfilter_amd.timeseries import models

    t_len = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    n_state = 8192
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

    rows = {}
    for n_theta in (1024, 512, 256, 128):
        best, rej = None, 0
        for rep in range(4):
            filt = APF(build, n_state, proposal=proposals.LinearGaussianObservations(), seed=2024 + rep)
            alg = SMC2(filt, n_theta, priors, threshold=0.2, device=device, dtype=dtype, seed=rep,
                       **({"block": int(os.environ["SMC2_BLOCK"])} if "SMC2_BLOCK" in os.environ else {}))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            alg.fit(y)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if rep and (best is None or dt < best):  # (rep 0 pays the plan / graph set-up)
                best, rej = dt, len(alg._kernel.acceptance_history)
        rows[n_theta] = (best, rej)
        print(f"theta-particles {n_theta:5d}: fit {1e3 * best:7.1f} ms, PMMH moves {rej}", flush=True)

    t1, rej1 = rows[1024]
    blocks = math.ceil(t_len / 16)
    print(f"\nmodel: {blocks} blocks of 16 observations; per rejuvenation an all-gather of theta + weights and an all-to-all of "
          f"<= the rank's filters' states ({n_state} x (4 + 4 + 4) B per filter, state dim 1)")
    for n in (2, 4, 8):
        t_rank, _ = rows[1024 // n]
        moved_bytes = (1024 // n) * n_state * 12 * (n - 1) / n  # expected share of a rank's ancestors owned elsewhere
        coll = blocks * ALL_GATHER_US * 1e-6 + rej1 * (2 * ALL_GATHER_US * 1e-6 + moved_bytes / (XGMI_GBS * 1e9))
        t_n = t_rank + coll
        print(f"N = {n}: rank compute {1e3 * t_rank:6.1f} ms + collectives {1e3 * coll:5.2f} ms = {1e3 * t_n:6.1f} ms  ->  speed-up "
              f"{t1 / t_n:4.2f}x, efficiency {t1 / (n * t_n):4.2f}   ({1024 * n_state * t_len / t_n:.3e} particle-steps/s)")


if __name__ == "__main__":
    main()
