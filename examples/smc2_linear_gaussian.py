#!/usr/bin/env python
"""SMC^2 on a linear-Gaussian state-space model with ``pyfilter_amd.inference.SMC2`` (the reference's
``pyfilter.inference.sequential.SMC2``: ``smc2.py:53-65``, ``kernels/mh.py:52-140``): theta-particles live on the filter's
batch dimension, every observation is one fused ``filter()`` move for all of them, and an ESS-triggered rejuvenation
resamples whole filters (``FilterResult.resample`` -> ``pf_columns_gather``), proposes from the Gaussian fitted to the
theta-particles, re-filters the data seen so far (one ``batch_filter`` call) and swaps the accepted filters in
(``FilterResult.exchange`` -> ``pf_columns_exchange``).  Parameters are updated in place - the filters read them live.

    x_t = beta * x_{t-1} + sigma * eps_t,     y_t = x_t + 0.3 * nu_t,     priors: beta ~ U(0, 1), sigma ~ U(0.05, 1)

Usage: python examples/smc2_linear_gaussian.py [n_theta] [n_state] [T]
       torchrun --nproc-per-node 8 examples/smc2_linear_gaussian.py 1024 8192 500      (theta-particles sharded)
"""
import os
import sys

import torch
from torch.distributions import Uniform

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from pyfilter_amd import timeseries as ts  # noqa: E402
from pyfilter_amd.filters.particle import APF, proposals  # noqa: E402
from pyfilter_amd.inference import SMC2  # noqa: E402
from pyfilter_amd.timeseries import models  # noqa: E402

PRIORS = {"beta": Uniform(0.0, 1.0), "sigma": Uniform(0.05, 1.0)}


def build_model(theta):
    """theta["beta"], theta["sigma"]: ``(B,)`` tensors the model keeps by reference."""
    dev = theta["beta"].device
    t = lambda v: torch.tensor(v, device=dev)  # noqa: E731
    return ts.LinearStateSpaceModel(models.AR(t(0.0), theta["beta"], theta["sigma"]), (t(1.0), t(0.3)))


def smc2(y: torch.Tensor, n_theta=256, n_state=2048, ess_frac=0.5, seed=0, verbose=False, block=1, **kernel_kwargs):
    """``block = 1``: observation by observation (``step``, the online use); ``block > 1``: ``fit`` with the filters running
    that many observations ahead of the rejuvenation test (one host decision point per block)."""
    filt = APF(build_model, n_state, proposal=proposals.LinearGaussianObservations(), seed=seed)
    alg = SMC2(filt, n_theta, PRIORS, threshold=ess_frac, device=y.device, seed=seed, **kernel_kwargs)
    state = alg.fit(y, block=block) if block > 1 else alg.initialize()
    for t, yt in enumerate(y if block <= 1 else []):
        before = len(alg._kernel.acceptance_history)
        state = alg.step(yt, state)
        if verbose and len(alg._kernel.acceptance_history) > before and alg.shard.rank == 0:
            print(f"t={t:3d}  rejuvenated, acceptance {alg._kernel.acceptance_history[-1]:.2f}, "
                  f"posterior mean {alg.posterior_mean(state).tolist()}, state particles {filt.particles[0]}")
    return dict(theta=alg.theta, weights=state.normalized_weights(), mean=alg.posterior_mean(state),
                moves=len(alg._kernel.acceptance_history), increases=alg._kernel._increases,
                loglikelihood=state.filter_state.loglikelihood, state_particles=filt.particles[0])


if __name__ == "__main__":
    n_theta = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    n_state = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    if int(os.environ.get("WORLD_SIZE", 1)) > 1:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
        torch.distributed.init_process_group("nccl")
    g = torch.Generator().manual_seed(1)
    beta, sigma, x, ys = 0.8, 0.4, 0.0, []
    for _ in range(T):
        x = beta * x + sigma * torch.randn((), generator=g).item()
        ys.append(x + 0.3 * torch.randn((), generator=g).item())
    out = smc2(torch.tensor(ys, device="cuda"), n_theta, n_state, verbose=True, block=int(os.environ.get("SMC2_BLOCK", 1)))
    if int(os.environ.get("RANK", 0)) == 0:
        print("posterior mean (beta, sigma):", out["mean"].tolist(), " truth:", (beta, sigma), " rejuvenations:", out["moves"])
