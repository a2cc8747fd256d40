#!/usr/bin/env python
"""Host wall time per building block of an SMC^2 fit at the reference's operating point (development tool): 1 000 theta x 400
state particles, T = 500, the OU model of tests/inference/models.py.  Wraps the driver's functions with perf_counter (no
synchronisation added) and prints the totals per fit.  Usage: python tools/smc2_phases.py"""
import os, sys, time, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _env  # noqa: E402  (tools/_env.py: PF_AMD_LIB / PF_* of this process -> the package's explicit switches)

_env.setup()
from torch.distributions import Exponential, LogNormal, Normal
from pyfilter_amd import timeseries as ts
from pyfilter_amd.filters.particle import APF, proposals
from pyfilter_amd.inference import SMC2
from pyfilter_amd.inference import smc2 as S, pmmh as P, parameters as PA, utils as U
from pyfilter_amd.timeseries import models
device, dtype = torch.device("cuda"), torch.float32
g = torch.Generator().manual_seed(123)
x, ys = 0.0, []
for _ in range(500):
    x = x * math.exp(-0.025) + 0.05 * math.sqrt((1 - math.exp(-0.05)) / 0.05) * torch.randn((), generator=g).item()
    ys.append(x + 0.05 * torch.randn((), generator=g).item())
y = torch.tensor(ys, dtype=dtype, device=device)
priors = {"kappa": Exponential(10.0), "gamma": Normal(0.0, 1.0), "sigma": LogNormal(-2.0, 1.0)}
# the observation constants live on the device ONCE: created inside the builder they would be two pageable host -> device
# copies - two waits for the device - per model build, and PMMH rebuilds the model at every move
obs_a, obs_s = torch.tensor(1.0, dtype=dtype, device=device), torch.tensor(0.05, dtype=dtype, device=device)

def build(theta):
    return ts.LinearStateSpaceModel(models.OrnsteinUhlenbeck(theta["kappa"], theta["gamma"], theta["sigma"], dt=1.0), (obs_a, obs_s))
acc = {}
def wrap(obj, name, key=None):
    f = getattr(obj, name); key = key or name
    def gfun(*a, **k):
        t1 = time.perf_counter()
        try: return f(*a, **k)
        finally: acc[key] = acc.get(key, 0.0) + time.perf_counter() - t1
    setattr(obj, name, gfun)
wrap(P, "run_pmmh"); S.run_pmmh = P.run_pmmh
wrap(S, "_take_filters")
wrap(PA.ThetaParticles, "resample", "theta.resample"); wrap(PA.ThetaParticles, "eval_priors"); wrap(PA.ThetaParticles, "unstack_parameters"); wrap(PA.ThetaParticles, "stack_parameters"); wrap(PA.ThetaParticles, "like"); wrap(PA.ThetaParticles, "exchange", "theta.exchange")
wrap(P.SymmetricMH, "build", "proposal.build")
wrap(APF, "copy", "filter.copy"); wrap(APF, "initialize_model"); wrap(APF, "batch_filter"); wrap(APF, "filter_block")
from pyfilter_amd.filters.result import FilterResult
wrap(FilterResult, "exchange", "result.exchange"); wrap(FilterResult, "resample", "result.resample")
wrap(S.ParticleMetropolisHastings, "update")
for rep in range(3):
    acc.clear()
    filt = APF(build, 400, proposal=proposals.LinearGaussianObservations(), seed=2024 + rep)
    alg = SMC2(filt, 1000, priors, threshold=0.2, device=device, dtype=dtype, seed=rep)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    alg.fit(y); torch.cuda.synchronize()
    print("fit ms", round(1e3 * (time.perf_counter() - t0), 1), {k: round(1e3 * v, 2) for k, v in sorted(acc.items(), key=lambda kv: -kv[1])})
