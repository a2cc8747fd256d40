#!/usr/bin/env python
"""SMC^2 on a linear-Gaussian state-space model, written against the hot path only (the moves
either side of the particle filter).  It is an *example*, not a port of ``pyfilter.inference.sequential.SMC2``
(``smc2.py:53-65``, ``kernels/mh.py:52-140``): theta-particles live on the filter's batch dimension, every observation is
one fused ``filter()`` move for all of them, and an ESS-triggered rejuvenation step resamples the theta-particles
(``FilterResult.resample`` -> ``pf_columns_gather``), re-filters the data seen so far for random-walk proposals (one
``batch_filter`` call, a replayed hipGraph) and swaps the accepted ones in (``FilterResult.exchange`` ->
``pf_columns_exchange``).  Parameters are updated in place - the filters read them live.

    x_t = beta * x_{t-1} + sigma * eps_t,     y_t = x_t + 0.3 * nu_t,     priors: beta ~ U(0, 1), sigma ~ U(0.05, 1)

Usage: python examples/smc2_linear_gaussian.py [n_theta] [n_state] [T]
"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from pyfilter_amd import resampling, timeseries as ts  # noqa: E402
from pyfilter_amd.filters.particle import APF, proposals  # noqa: E402
from pyfilter_amd.timeseries import models  # noqa: E402

LO = torch.tensor([0.0, 0.05])
HI = torch.tensor([1.0, 1.0])


def build_filter(theta: torch.Tensor, n_state: int, seed: int):
    """theta (n_theta, 2) = (beta, sigma) - the filter keeps views of its columns, so in-place edits are seen live."""
    dev = theta.device
    t = lambda v: torch.tensor(v, device=dev)  # noqa: E731
    ssm = ts.LinearStateSpaceModel(models.AR(t(0.0), theta[:, 0], theta[:, 1]), (t(1.0), t(0.3)))
    f = APF(ssm, n_state, proposal=proposals.LinearGaussianObservations(), seed=seed)
    f.set_batch_shape(torch.Size([theta.shape[0]]))
    return f


def log_prior(theta):
    inside = ((theta > LO.to(theta.device)) & (theta < HI.to(theta.device))).all(dim=1)
    return torch.where(inside, torch.zeros_like(theta[:, 0]), torch.full_like(theta[:, 0], -math.inf))


def smc2(y: torch.Tensor, n_theta=256, n_state=2048, ess_frac=0.5, seed=0, verbose=False):
    dev = y.device
    gen = torch.Generator(device=dev).manual_seed(seed)
    theta = LO.to(dev) + (HI - LO).to(dev) * torch.rand(n_theta, 2, device=dev, generator=gen)
    # the theta tensor is the storage of the model parameters: column views go into the model
    filt = build_filter(theta, n_state, seed)
    state = filt.initialize()
    result = filt.initialize_with_result(state)
    logw = torch.zeros(n_theta, device=dev)          # theta log-weights
    moves = 0
    for t in range(y.shape[0]):
        state = filt.filter(y[t], state, result=result)
        logw = logw + state.get_loglikelihood()
        w = torch.softmax(logw, 0)
        if 1.0 / (w * w).sum() < ess_frac * n_theta and t + 1 < y.shape[0]:
            # ---- resample theta-particles (and the whole filter states with them) ----------------------------------
            idx = resampling.systematic(w, normalized=True)
            theta.copy_(theta[idx])
            result.resample(idx)
            state = result.latest_state
            logw.zero_()
            # ---- one PMMH move: random-walk proposal, re-filter y[:t+1], accept / reject ------------------------------
            std = theta.std(dim=0)
            prop = theta + 1.5 * std * torch.randn(theta.shape, device=dev, generator=gen) / math.sqrt(2.0)
            ok = torch.isfinite(log_prior(prop))
            prop = torch.where(ok[:, None], prop, theta)     # proposals outside the prior support are rejected anyway
            f_new = build_filter(prop, n_state, seed + 1000 + t)
            r_new = f_new.batch_filter(y[: t + 1], bar=False)
            log_acc = r_new.loglikelihood - result.loglikelihood + log_prior(prop) - log_prior(theta)
            accept = ok & (torch.rand(n_theta, device=dev, generator=gen).log() < log_acc)
            result.exchange(r_new, accept)
            theta[accept] = prop[accept]                     # in place: `filt` reads the new parameters on its next move
            state = result.latest_state
            moves += 1
            if verbose:
                print(f"t={t:3d}  rejuvenated, acceptance {accept.float().mean().item():.2f}, "
                      f"theta mean {theta.mean(0).tolist()}")
    w = torch.softmax(logw, 0)
    mean = (w[:, None] * theta).sum(0)
    return dict(theta=theta, weights=w, mean=mean, moves=moves, loglikelihood=result.loglikelihood)


if __name__ == "__main__":
    n_theta = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    n_state = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    g = torch.Generator().manual_seed(1)
    beta, sigma, x, ys = 0.8, 0.4, 0.0, []
    for _ in range(T):
        x = beta * x + sigma * torch.randn((), generator=g).item()
        ys.append(x + 0.3 * torch.randn((), generator=g).item())
    out = smc2(torch.tensor(ys, device="cuda"), n_theta, n_state, verbose=True)
    print("posterior mean (beta, sigma):", out["mean"].tolist(), " truth:", (beta, sigma), " rejuvenations:", out["moves"])
