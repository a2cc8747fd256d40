import cProfile, pstats, sys, os, math, torch
sys.path.insert(0, os.getcwd())
from torch.distributions import Exponential, LogNormal, Normal
from pyfilter_amd import timeseries as ts
from pyfilter_amd.filters.particle import APF, proposals
from pyfilter_amd.inference import SMC2
from pyfilter_amd.timeseries import models
device, dtype = torch.device("cuda"), torch.float32
g = torch.Generator().manual_seed(123)
x, ys = 0.0, []
for _ in range(500):
    x = x * math.exp(-0.025) + 0.05 * math.sqrt((1 - math.exp(-0.05)) / 0.05) * torch.randn((), generator=g).item()
    ys.append(x + 0.05 * torch.randn((), generator=g).item())
y = torch.tensor(ys, dtype=dtype, device=device)
priors = {"kappa": Exponential(10.0), "gamma": Normal(0.0, 1.0), "sigma": LogNormal(-2.0, 1.0)}
def build(theta):
    t = lambda v: torch.tensor(v, dtype=dtype, device=device)
    return ts.LinearStateSpaceModel(models.OrnsteinUhlenbeck(theta["kappa"], theta["gamma"], theta["sigma"], dt=1.0), (t(1.0), t(0.05)))
for rep in range(2):
    filt = APF(build, 8192, proposal=proposals.LinearGaussianObservations(), seed=2024 + rep)
    alg = SMC2(filt, 1024, priors, threshold=0.2, device=device, dtype=dtype, seed=rep)
    state = alg.initialize()
    for t in range(300):
        state = alg.step(y[t], state)
    torch.cuda.synchronize()
    if rep == 1:
        pr = cProfile.Profile()
        pr.enable()
        for t in range(300, 500):
            state = alg.step(y[t], state)
        torch.cuda.synchronize()
        pr.disable()
        st = pstats.Stats(pr)
        st.sort_stats("tottime").print_stats(28)
        st.sort_stats("cumulative").print_stats(40)
