#!/usr/bin/env python
"""Where the HOST spends an SMC^2 fit (development tool): cProfile of one ``SMC2.fit`` at 128 theta-particles x 8 192 state
particles, T = 500 - the per-rank job of an 8-GPU run, which is host-bound (tools/smc2_scaling_model.py).
Usage: python tools/smc2_host_profile.py [n_theta] [n_state]"""
import cProfile
import math
import os
import pstats
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

    n_theta = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    n_state = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    kw = {"block": int(os.environ["SMC2_BLOCK"])} if "SMC2_BLOCK" in os.environ else {}
    device, dtype, t_len = torch.device("cuda"), torch.float32, 500
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

    def fit(seed):
        filt = APF(build, n_state, proposal=proposals.LinearGaussianObservations(), seed=2024 + seed)
        alg = SMC2(filt, n_theta, priors, threshold=0.2, device=device, dtype=dtype, seed=seed, **kw)
        alg.fit(y)
        torch.cuda.synchronize()
        return alg

    fit(0)
    t0 = time.perf_counter()
    fit(1)
    print(f"fit at {n_theta} theta x {n_state}: {1e3 * (time.perf_counter() - t0):.1f} ms")
    pr = cProfile.Profile()
    pr.enable()
    fit(2)
    pr.disable()
    st = pstats.Stats(pr)
    rows = [(tt, ct, nc, f"{os.path.basename(fn)}:{ln}({name})") for (fn, ln, name), (cc, nc, tt, ct, _) in st.stats.items()]
    for title, key in (("cumulative", 1), ("own", 0)):  # (microseconds: pstats prints milliseconds, too coarse for a 12 ms fit)
        print(f"--- top 45 by {title} time: own us, cumulative us, calls")
        for tt, ct, nc, what in sorted(rows, key=lambda r: -r[key])[:45]:
            print(f"{1e6 * tt:9.0f} {1e6 * ct:9.0f} {nc:6d}  {what}")


if __name__ == "__main__":
    main()
