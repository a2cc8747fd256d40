"""The one linear-Gaussian shape that stays on the TORCH route - a scalar state under a vector observation (D = 1, O > 1:
``proposals/utils.py:243-245`` ``hidden_is_1d`` with a matrix observation; the fused kernels take ``D > 1 or O == 1``) - against
fixtures of the unmodified reference (``tests/golden/lg1d_o2_*``, ``oracle/make_golden.py``).

The route draws from ``torch.distributions`` (torch's generator: no tape to inject), so the check is teacher-forced, move by move
on the reference's own recorded states: from the reference's state ``t`` (resampled with the reference's recorded uniform by the
oracle's ``sisr_predict``) this package's proposal - ``LinearGaussianObservations`` through ``_ObservationUpdate`` (innovation
form), ``Bootstrap`` through the model's densities - must (i) place the optimal proposal where the reference's next particles are
(``mean + std z`` with the recorded normals ``z`` reproduces ``step_x[t]``) and (ii) give the reference's weights for those
particles, NaN observation (propagate only) included.  The route is device-agnostic torch code: run on the CPU here and on the
GPU under ``-m gpu``; the run as a whole is checked statistically against the reference's log-likelihood."""
import math

import pytest
import torch

from oracle import cpu_ref
from oracle.cases import CASE_BY_NAME, build_spec
from tests.helpers import build_ssm_from_case, load_golden

NAMES = ["lg1d_o2_sisr_lgo", "lg1d_o2_sisr_boot"]
DEVICES = ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)]


@pytest.mark.parametrize("device", DEVICES)
@pytest.mark.parametrize("name", NAMES)
def test_scalar_state_under_a_vector_observation_teacher_forced(name, device):
    from pyfilter_amd.filters.particle import proposals
    from pyfilter_amd.filters.particle.proposals.linear import _ObservationUpdate
    from pyfilter_amd.timeseries import TimeseriesState

    case = CASE_BY_NAME[name]
    g = load_golden(name, "f64")
    spec = build_spec(case, torch.float64)
    ssm = build_ssm_from_case(case, torch.float64, device)
    assert getattr(ssm, "kernel_kind", None) is None, "D = 1 / O = 2 is expected on the torch route"
    prop = {"lgo": proposals.LinearGaussianObservations, "bootstrap": proposals.Bootstrap}[case["proposal"]]()
    prop.set_model(ssm)
    thr = case["ess_threshold"] * case["N"]
    x, w = g["x0"].double(), torch.zeros_like(g["x0"].double())
    prev = torch.arange(case["N"]).unsqueeze(-1).expand(case["N"], case["B"])
    for t in range(case["T"]):
        y, z, u = g["y"][t].double(), g["z_tape"][t].double(), g["u_tape"][t].double()
        xr, wr, W, idx, _ = cpu_ref.sisr_predict(spec, x, w, prev, u, thr)  # the reference's resampling of its own state
        assert torch.equal(idx, g["step_idx"][t]), f"move {t}: the fixture's ancestors"
        ts_state = TimeseriesState(t, xr.to(device), ssm.hidden.event_shape)
        mean, scale = ssm.hidden.mean_scale(ts_state)
        x_ref = g["step_x"][t].double().to(device)
        if torch.isnan(y).all():  # propagate only, weights carried (filters/base.py:212, particle/state.py:38-42)
            torch.testing.assert_close(mean + scale * z.to(device), x_ref, rtol=1e-9, atol=1e-12)
            w_new = wr
        else:
            yd = y.to(device)
            if case["proposal"] == "lgo":
                kernel = _ObservationUpdate(ssm, scale).posterior(yd, mean)
                torch.testing.assert_close(kernel.mean + kernel.stddev * z.to(device), x_ref, rtol=1e-9, atol=1e-12)
                wi = prop._weight_with_kernel(yd, ssm.hidden.build_density(ts_state), ts_state.copy(values=mean).propagate_from(values=x_ref), kernel)
            else:
                torch.testing.assert_close(mean + scale * z.to(device), x_ref, rtol=1e-9, atol=1e-12)
                wi = ssm.build_density(ts_state.copy(values=mean).propagate_from(values=x_ref)).log_prob(yd)
            w_new = wi.cpu() + wr
        torch.testing.assert_close(w_new, g["step_w"][t].double(), rtol=1e-9, atol=1e-9)
        x, w, prev = g["step_x"][t].double(), g["step_w"][t].double(), idx


@pytest.mark.gpu
@pytest.mark.parametrize("device", ["cuda"])
def test_scalar_state_under_a_vector_observation_end_to_end(device):
    """The whole filter on its own draws: 64 independent runs of SISR + LinearGaussianObservations at the fixture's size must
    scatter around the reference's log-likelihood (the reference's value is one such run: within 4 standard deviations of the
    runs' mean) and report finite moments."""
    from pyfilter_amd.filters.particle import SISR, proposals

    name = "lg1d_o2_sisr_lgo"
    case = CASE_BY_NAME[name]
    g = load_golden(name, "f64")
    ssm = build_ssm_from_case(case, torch.float64, device)
    torch.manual_seed(7)
    filt = SISR(ssm, case["N"], proposal=proposals.LinearGaussianObservations(), ess_threshold=case["ess_threshold"], seed=3)
    filt.set_batch_shape(torch.Size([64]))
    res = filt.batch_filter(g["y"].double().to(device), bar=False)
    ll = res.loglikelihood.cpu()
    assert torch.isfinite(ll).all() and torch.isfinite(res.filter_means).all()
    ref = g["loglikelihood"].double()
    assert (ref - ll.mean()).abs().max() < 4.0 * ll.std() + 1e-3, (ref.tolist(), ll.mean().item(), ll.std().item())
    assert ll.std() < 1.0
