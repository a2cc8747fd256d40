"""User-defined affine models - the reference's plug-in seam, a ``mean_scale`` lambda (``/root/reference/README.md:44-67``,
``examples/lorenz.ipynb:53-58``) - on the FUSED route: the callable is evaluated once per step with PyTorch-ROCm ops into
(loc, scale) planes and the fused kernels (``PF_HID_USER_AFFINE``) gather them at the ancestors and do everything else of
the step.  The lambda-defined twins of the golden cases' models must reproduce the reference's fixtures exactly like the
built-in kinds do: float64, identical draws, identical ancestors, means / log-likelihood to 1e-9."""
import math

import pytest
import torch

from oracle.cases import CASE_BY_NAME
from tests.conftest import both_routes
from tests.helpers import load_golden

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _lambda_ssm(model, b, dtype):
    """The golden cases' models (tests/helpers.py::build_ssm_from_case) written the reference's way: python callables."""
    from torch.distributions import Independent, Normal

    from pyfilter_amd import timeseries as ts

    t = lambda v: torch.tensor(v, dtype=dtype, device=DEV)  # noqa: E731
    if model == "lg1d":
        hidden = ts.AffineProcess(lambda x, a, bb, s: (a + bb * x.value, s), (t(0.0), t(0.99), t(0.05)), Normal(t(0.0), t(1.0)),
                                  lambda a, bb, s: Normal(t(0.0), t(0.05)))
        return ts.LinearStateSpaceModel(hidden, (t(1.0), t(0.15)))
    if model == "sine":
        hidden = ts.AffineEulerMaruyama(lambda x, gm, s: (torch.sin(x.value - gm), s), (t(0.0), t(1.0)),
                                        Normal(t(0.0), t(math.sqrt(0.1))), 0.1, lambda gm, s: Normal(t(0.0), t(1.0)))
        return ts.LinearStateSpaceModel(hidden, (t(1.0), t(0.1)))
    if model == "lorenz":
        def f(x, s, r, bb, sigma):  # examples/lorenz.ipynb:53-58
            v = x.value
            return torch.stack((-s * (v[..., 0] - v[..., 1]), r * v[..., 0] - v[..., 1] - v[..., 0] * v[..., 2],
                                v[..., 0] * v[..., 1] - bb * v[..., 2]), dim=-1), sigma

        inc = Independent(Normal(t(0.0), t(math.sqrt(0.01))).expand(torch.Size([3])), 1)
        init = lambda *_: Independent(Normal(t([-5.91652, -5.52332, 24.5723]), t([math.sqrt(10.0)] * 3)), 1)  # noqa: E731
        hidden = ts.AffineEulerMaruyama(f, (t(10.0), t(28.0), t(8.0 / 3.0), t(1.0)), inc, 0.01, init)
        a = t([[0.8, 0.0, 0.0], [0.0, 0.0, 0.8]])
        return ts.LinearStateSpaceModel(hidden, (a, t([0.0]), t([math.sqrt(0.1)])), torch.Size([2]))
    if model == "ou_batched":
        kappa, gamma = t([0.025 * (i + 1) for i in range(b)]), t([0.1 * i for i in range(b)])
        sigma = t([0.05 + 0.01 * i for i in range(b)])

        def ou(x, k, gm, s):
            return gm + (x.value - gm) * torch.exp(-k), s * torch.sqrt((1.0 - torch.exp(-2.0 * k)) / (2.0 * k))

        hidden = ts.AffineProcess(ou, (kappa, gamma, sigma), Normal(t(0.0), t(1.0)), lambda *_: Normal(t(0.0), t(0.1)))
        return ts.LinearStateSpaceModel(hidden, (t(1.0), t(0.05)))
    raise KeyError(model)


NAMES = ["lg1d_sisr_boot", "lg1d_apf_lgo", "sine_apf_lgo", "sine_sisr_lgo", "sine_apf_boot_nan", "sine_sisr_boot_nan",
         "lorenz_sisr_boot", "lorenz_apf_lgo", "ou_sisr_lgo_theta", "ou_apf_boot_theta"]


@both_routes
@pytest.mark.parametrize("loop", ["run", "moves"])
@pytest.mark.parametrize("name", NAMES)
def test_lambda_defined_models_reproduce_the_reference_fixtures(name, loop, kernel_route, monkeypatch):
    """``loop``: "run" - ``batch_filter``'s own loop over the moves (callable -> planes -> one fused run per move, on the
    plan's buffers); "moves" - the reference's driver loop over ``filter()`` (one fused single-step move per call)."""
    if loop == "moves":
        from pyfilter_amd.hints import HINTS

        monkeypatch.setattr(HINTS, "fused_batch", False)
    from pyfilter_amd import ops
    from pyfilter_amd.filters.particle import APF, SISR, proposals
    from pyfilter_amd.filters.particle.state import ParticleFilterCorrection
    from pyfilter_amd.timeseries import TimeseriesState

    case, g = CASE_BY_NAME[name], load_golden(name, "f64")
    dtype, n, b = torch.float64, case["N"], case["B"]
    ssm = _lambda_ssm(case["model"], b, dtype)
    assert ssm.kernel_kind is not None and ssm.kernel_kind.is_user
    prop = {"bootstrap": proposals.Bootstrap, "lgo": proposals.LinearGaussianObservations}[case["proposal"]]()
    filt = {"sisr": SISR, "apf": APF}[case["filter"]](ssm, n, proposal=prop, ess_threshold=case["ess_threshold"])
    filt.set_batch_shape(torch.Size([b]))
    filt.set_tape(z=g["z_tape"].to(dtype), u=g["u_tape"].to(dtype))
    x0 = g["x0"].to(DEV)
    state = ParticleFilterCorrection(TimeseriesState(0, x0, ssm.hidden.event_shape), torch.zeros((n, b), dtype=dtype, device=DEV),
                                     torch.zeros(b, dtype=dtype, device=DEV), torch.arange(n, device=DEV).unsqueeze(-1).expand(n, b))
    res = filt.batch_filter(g["y"].to(DEV), bar=False, init_state=state)
    torch.cuda.synchronize()
    rec = ops.debug_launch_trace(1)[-1]
    assert (rec["SPEC"] == 9) if kernel_route == "column" else (rec["MK"] == 3 and rec["FAST"] == 0), rec  # the fused kernels ran
    tol = dict(rtol=1e-9, atol=1e-11)
    torch.testing.assert_close(res.filter_means.cpu(), g["filter_means"], **tol)
    torch.testing.assert_close(res.filter_variance.cpu(), g["filter_variance"], rtol=1e-8, atol=1e-11)
    torch.testing.assert_close(res.loglikelihood.cpu(), g["loglikelihood"], **tol)
    last = res.latest_state
    assert torch.equal(last.previous_indices.cpu(), g["step_idx"][-1]), "final ancestors differ"
    torch.testing.assert_close(last.timeseries_state.value.cpu(), g["step_x"][-1], **tol)
    torch.testing.assert_close(last.weights.cpu(), g["step_w"][-1], equal_nan=True, **tol)


def test_readme_lambda_model_at_a_million_particles_is_fused():
    """The README's sine diffusion as a lambda (README.md:44-67) at 2^20 particles, float32, Philox draws: every step is the
    user's torch ops + the fused kernels (reduce, step, bookkeeping) - and the filter agrees with the built-in kind."""
    from torch.distributions import Normal

    from pyfilter_amd import ops, timeseries as ts
    from pyfilter_amd.filters.particle import APF, proposals
    from pyfilter_amd.timeseries import models

    t = lambda v: torch.tensor(v, dtype=torch.float32, device=DEV)  # noqa: E731
    user = ts.AffineEulerMaruyama(lambda x, gm, s: (torch.sin(x.value - gm), s), (t(0.0), t(1.0)), Normal(t(0.0), t(math.sqrt(0.1))),
                                  0.1, lambda gm, s: Normal(t(0.0), t(1.0)))
    ssm_user = ts.LinearStateSpaceModel(user, (t(1.0), t(0.1)))
    ssm_builtin = ts.LinearStateSpaceModel(models.SineDiffusion(t(0.0), t(1.0), dt=0.1), (t(1.0), t(0.1)))
    g = torch.Generator().manual_seed(4)
    x, ys = 0.3, []
    for _ in range(40):
        x = x + math.sin(x) * 0.1 + math.sqrt(0.1) * torch.randn((), generator=g).item()
        ys.append(x + 0.1 * torch.randn((), generator=g).item())
    y = torch.tensor(ys, dtype=torch.float32, device=DEV)
    torch.manual_seed(0)
    n = 1 << 20
    ru = APF(ssm_user, n, proposal=proposals.LinearGaussianObservations()).batch_filter(y, bar=False)
    torch.cuda.synchronize()
    rec = ops.debug_launch_trace(1)[-1]
    assert rec["MK"] == 3 and rec["FAST"] == 0 and rec["SPEC"] != 9
    rb = APF(ssm_builtin, n, proposal=proposals.LinearGaussianObservations()).batch_filter(y, bar=False)
    assert ru.filter_means.shape == rb.filter_means.shape == (41, 1)
    assert (ru.filter_means[1:] - rb.filter_means[1:]).abs().max().item() < 5e-3  # two Monte-Carlo runs of 2^20 particles
    assert abs((ru.loglikelihood - rb.loglikelihood).item()) < 0.05
