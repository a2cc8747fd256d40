import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_user_models_gpu import _lambda_ssm
from pyfilter_amd.filters import particle as pfm
from pyfilter_amd.filters.particle import proposals
DEV = "cuda"
def run(graphed, n=4096, b=3, reps=3, cls="APF", prop="lgo"):
    ssm = _lambda_ssm("sine", b, torch.float32)
    ssm.hidden.graph_callable = graphed
    y = (0.1 * torch.randn((12,), generator=torch.Generator().manual_seed(5))).cumsum(0).to(DEV)
    p = {"bootstrap": proposals.Bootstrap, "lgo": proposals.LinearGaussianObservations}[prop]()
    filt = getattr(pfm, cls)(ssm, n, proposal=p, seed=99)
    if b > 1: filt.set_batch_shape(torch.Size([b]))
    out = []
    for r in range(reps):
        res = filt.batch_filter(y, bar=False)
        out.append((res.filter_means.cpu(), res.loglikelihood.cpu(), res.latest_state.previous_indices.cpu(), filt._last_run["seed_eff"]))
    return out
for cls, prop in (("APF", "lgo"), ("SISR", "bootstrap")):
    a, b_, c = run(False, cls=cls, prop=prop), run(False, cls=cls, prop=prop), run(True, cls=cls, prop=prop)
    for r in range(3):
        print(cls, prop, "run", r, "seeds", a[r][3], b_[r][3], c[r][3], "eager-vs-eager: idx equal", torch.equal(a[r][2], b_[r][2]), "means maxdiff", float((a[r][0]-b_[r][0]).abs().max()),
              "| eager-vs-graph: idx equal", torch.equal(a[r][2], c[r][2]), "means maxdiff", float((a[r][0]-c[r][0]).abs().max()), "ll", a[r][1].tolist(), c[r][1].tolist())
