"""f2 pinned against the reference, GPU leg: ``pyfilter_amd.inference`` on the HIP filters
(fp64, every particle-level and theta-level draw injected) reproduces the event logs recorded from the unmodified
reference's ``SMC2._step`` / ``ParticleMetropolisHastings.update`` / ``_increase_states`` / ``run_pmmh`` - the online
fused ``filter()`` moves, the fused ``batch_filter`` re-runs of every PMMH move (model rebuilt from theta* each time), the
whole-filter ``resample`` / ``exchange`` kernels, the particle doubling: log-likelihood increments, theta-weights, ESS,
theta ancestors, Gaussian proposal, theta*, log acceptance probabilities, accepted masks, theta and filter moments after
every update."""
import pytest
import torch

from tests.replay import taped
from tests.test_inference_reference_cpu import SMC2_CASES, replay_pmmh, replay_smc2

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["theta_kernels", "theta_torch"], autouse=True)
def theta_route(request):
    """Every replay runs on both theta routes: the move's theta arithmetic in ``pf_theta_fit / _propose / _accept``
    (``csrc/pf_theta.hpp`` - taken for the scalar Exponential / Normal / LogNormal priors of these cases) and in
    ``torch.distributions``."""
    from pyfilter_amd.hints import HINTS

    HINTS.theta_kernels = request.param == "theta_kernels"
    yield request.param
    HINTS.theta_kernels = True


def _hip_filter(cursor, n):
    from pyfilter_amd import timeseries as ts
    from pyfilter_amd.filters.particle import APF, proposals
    from pyfilter_amd.timeseries import models

    def build(theta):  # tests/inference/models.py:22-33 (the OU process with its stationary initial distribution)
        t = lambda v: torch.tensor(v, dtype=torch.float64, device="cuda")  # noqa: E731
        return ts.LinearStateSpaceModel(models.OrnsteinUhlenbeck(theta["kappa"], theta["gamma"], theta["sigma"], dt=1.0), (t(1.0), t(0.05)))

    cls = taped(APF)
    cls.cursor = cursor
    return cls(build, n, proposal=proposals.LinearGaussianObservations())


@pytest.mark.parametrize("name", sorted(SMC2_CASES))
def test_smc2_on_the_hip_filters_replays_the_reference_event_log(name):
    updates, outcomes = replay_smc2(name, _hip_filter, "cuda", rtol=1e-7)
    assert updates == len(outcomes) >= 2
    if name.endswith("increase"):
        assert "increase" in outcomes and "done" in outcomes


def test_run_pmmh_random_walk_on_the_hip_filters_replays_the_reference_event_log():
    assert replay_pmmh("inference_pmmh_ou_rw", _hip_filter, "cuda", rtol=1e-7) == 5


def test_the_proposal_filter_draws_x0_from_the_proposed_theta():
    """``run_pmmh`` rebuilds the model from theta* (mcmc/utils.py:52-53) - an OU process derives its
    stationary initial scale sigma / sqrt(2 kappa) when it is built, so a proposal filter that kept the old model would
    start every re-run from the OLD theta's initial law."""
    from torch.distributions import Exponential, LogNormal, Normal

    from pyfilter_amd import timeseries as ts
    from pyfilter_amd.filters.particle import APF
    from pyfilter_amd.inference import SymmetricMH, ThetaParticles
    from pyfilter_amd.inference.pmmh import PMMHState, run_pmmh
    from pyfilter_amd.timeseries import models

    dev, dt = "cuda", torch.float64
    pri = {"kappa": Exponential(10.0), "gamma": Normal(0.0, 1.0), "sigma": LogNormal(-2.0, 1.0)}
    b = 8
    theta = ThetaParticles(pri, b, dev, dt).initialize_parameters(torch.Generator().manual_seed(1))

    def build(th):
        t = lambda v: torch.tensor(v, dtype=dt, device=dev)  # noqa: E731
        return ts.LinearStateSpaceModel(models.OrnsteinUhlenbeck(th["kappa"], th["gamma"], th["sigma"]), (t(1.0), t(0.05)))

    filt = APF(build, 20000)
    filt.set_batch_shape(torch.Size([b]))
    filt.initialize_model(theta)
    y = torch.zeros(3, dtype=dt, device=dev)
    state = PMMHState(filt.batch_filter(y, bar=False), theta.stack_parameters(True), 2)
    prop = SymmetricMH()
    kernel = prop.build(theta, state, filt, y)
    ptheta, pfilt = theta.like(), filt.copy()
    pfilt.initialize_model(ptheta)
    run_pmmh(theta, state, prop, kernel, pfilt, ptheta, y, torch.Size([b]), generator=torch.Generator().manual_seed(2))
    want = ptheta["sigma"] / torch.sqrt(2.0 * ptheta["kappa"])  # theta* (the proposal's parameters hold the proposed values)
    got = pfilt.initialize().timeseries_state.value.std(dim=0)
    torch.testing.assert_close(got, want, rtol=0.05, atol=0.0)
    torch.testing.assert_close(pfilt.ssm.hidden.init_scale.reshape(-1).expand(b), want, rtol=1e-12, atol=0.0)
