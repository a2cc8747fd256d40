#!/usr/bin/env python
"""Every device -> host synchronisation of one ``SMC2.fit`` (development tool): torch's sync debug mode prints a warning with
the Python stack of each; the count per call site is what is reported.  Usage: python tools/smc2_syncs.py [n_theta] [n_state]"""
import collections
import math
import os
import sys
import traceback
import warnings

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

    n_theta = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    n_state = int(sys.argv[2]) if len(sys.argv) > 2 else 400
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
        alg = SMC2(filt, n_theta, priors, threshold=0.2, device=device, dtype=dtype, seed=seed)
        alg.fit(y)
        torch.cuda.synchronize()
        return alg

    fit(0)
    sites = collections.Counter()

    def showwarning(message, category, filename, lineno, file=None, line=None):
        if "synchroniz" not in str(message):
            return
        stack = [f for f in traceback.extract_stack() if "/pyfilter_amd/" in f.filename]
        where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}({f.name})" for f in reversed(stack[-3:]))
        sites[where] += 1

    warnings.showwarning = showwarning
    warnings.simplefilter("always")
    torch.cuda.set_sync_debug_mode("warn")
    alg = fit(1)
    torch.cuda.set_sync_debug_mode("default")
    print(f"SMC2.fit {n_theta} theta x {n_state}, T = {t_len}: {sum(sites.values())} synchronising calls, "
          f"{len(alg._kernel.acceptance_history)} PMMH moves")
    for where, k in sites.most_common():
        print(f"  {k:4d}  {where}")


if __name__ == "__main__":
    main()
