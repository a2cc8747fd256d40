import sys, warnings, torch
sys.path.insert(0, ".")
from pyfilter_amd import resampling, timeseries as ts
from pyfilter_amd.filters.particle import APF, proposals
from pyfilter_amd.hints import HINTS
from pyfilter_amd.inference import SMC2
from pyfilter_amd.timeseries import models
from torch.distributions import Exponential, LogNormal, Normal
DEV = "cuda"
g = torch.Generator().manual_seed(3)
y = (0.05 * torch.randn(40, generator=g, dtype=torch.float64)).cumsum(0).to(DEV)
priors = {"kappa": Exponential(10.0), "gamma": Normal(0.0, 1.0), "sigma": LogNormal(-2.0, 1.0)}
obs_a, obs_s = torch.tensor(1.0, device=DEV, dtype=torch.float64), torch.tensor(0.05, device=DEV, dtype=torch.float64)
def build(theta):
    return ts.LinearStateSpaceModel(models.OrnsteinUhlenbeck(theta["kappa"], theta["gamma"], theta["sigma"], dt=1.0), (obs_a, obs_s))
def fit(cluster, patience, route=0):
    HINTS.cluster, HINTS.cluster_patience, HINTS.route = cluster, patience, route
    filt = APF(build, 4096, proposal=proposals.Bootstrap(), resampling=resampling.systematic, seed=5)
    alg = SMC2(filt, 16, priors, threshold=0.5, device=torch.device(DEV), dtype=torch.float64, seed=9)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        state = alg.initialize()
        for t in range(y.shape[0]):
            state = alg.step(y[t], state)
    torch.cuda.synchronize()
    return torch.stack(state.ess).cpu(), state.w.cpu(), state.filter_state.loglikelihood.cpu(), getattr(filt, "cluster_fallbacks", 0)
a = fit(False, 0)
a2 = fit(False, 0)
b = fit(True, 1)
c = fit(True, 0)
d = fit(False, 0, route=1)
for name, x in (("ref again", a2), ("patience 1", b), ("cluster ok", c), ("route 1", d)):
    de = (x[0] - a[0]).abs()
    print(name, "fallbacks", x[3], "ess maxdiff", float(de.max()), "first", int((de > 0).nonzero()[0]) if (de > 0).any() else -1,
          "w", float((x[1] - a[1]).abs().max()), "ll", float((x[2] - a[2]).abs().max()))
