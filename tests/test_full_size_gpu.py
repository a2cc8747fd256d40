"""Full-length runs at full size: every BASELINE configuration for its own T at its own N (BASELINE.json ``configs[1..4]``;
``bench.py`` times exactly these).  The step-level parity suites run T = 4 - 60 at these sizes; what only a long run can show -
moment pivots that drift, poison flags that stick, weights that degenerate, the seams of the launch sequence, a graph
replay that goes stale - is checked here through properties that do not need a CPU oracle of the same size:

* config 2 (sine diffusion, APF + LinearGaussianObservations, 2^20 particles, T = 250, float32): against the float64 run of
  the same filter on the same observations (filter means within Monte-Carlo error at every step, log-likelihood), against
  the ORACLE's float64 log-likelihood of the same seeded data (``bench.EXPECTED_LL``: ``tools/bench_reference_ll.py``, CPU), and
  the production instantiation ran at every step;
* config 5's filtering pass (1 024 theta x 8 192, OU, APF + LGO, T = 500): the exact Kalman log-likelihood and filter
  means of every theta-particle;
* config 3 (64 series x 65 536, stochastic volatility, APF + Bootstrap, T = 1 000): invariants + the float64 run of the
  first 8 series;
* config 4 (Lorenz-63, SISR + Bootstrap, multinomial, 2^22 particles, T = 200 of its 2 000): invariants + float64 / systematic
  runs within the spread of two independent float64 runs.

Whole file: about half a minute on one MI355X."""
import math

import pytest
import torch

from oracle import cpu_ref
from oracle.cases import build_spec, simulate
from pyfilter_amd import ops
from tests.helpers import build_ssm_from_case

pytestmark = pytest.mark.gpu
F32, F64 = torch.float32, torch.float64


def _trace(t_len):
    """The launch records of the run that just finished, oldest first (the library keeps the last ``ops.TRACE_LEN``)."""
    tr = ops.debug_launch_trace(t_len)
    assert len(tr) == min(t_len, ops.TRACE_LEN) and [r["step"] for r in tr] == list(range(t_len))[-len(tr):], tr[:2]
    return tr


def _sorted_in_range(idx, n):
    return bool((idx[1:] >= idx[:-1]).all()) and int(idx.min()) >= 0 and int(idx.max()) <= n - 1


def test_config2_full_length_float32_against_float64_and_the_oracle_likelihood():
    import bench

    n, t_len = 1 << 20, 250
    runs = {}
    for dtype, seed in ((F32, 2024), (F64, 7), (F64, 8)):
        filt, y, w = bench.build_problem("apf_lgo_1m", dtype, "cuda", 1, 0)
        assert w["N"] == n and w["T"] == t_len and y.shape == (t_len,)
        filt._seed = seed
        res = filt.batch_filter(y, bar=False)
        torch.cuda.synchronize()
        if dtype == F32:
            tr = _trace(t_len)
            # the headline instantiation (closed form, APF observed -> observed) at every step but the run's last
            assert all(r["tbytes"] == 4 and r["D"] == 1 and r["FAST"] == 1 and r["SPEC"] == 1 for r in tr[:-1]), tr[:3]
            res2 = filt.batch_filter(y, bar=False)  # (a second run: other draws - every run takes a fresh epoch)
            assert not torch.equal(res2.loglikelihood, res.loglikelihood)
            runs["f32b"] = res2
        runs[(dtype, seed)] = res
    r32, r64a, r64b = runs[(F32, 2024)], runs[(F64, 7)], runs[(F64, 8)]
    for r in runs.values():
        assert r.filter_means.shape == (t_len + 1, 1) and torch.isfinite(r.filter_means).all() and torch.isfinite(r.loglikelihood).all()
        assert torch.isfinite(r.filter_variance).all() and (r.filter_variance >= 0).all()
        assert _sorted_in_range(r.latest_state.previous_indices, n)
    # Monte-Carlo scale per step: the cloud's sqrt(var / N), floored by what two independent float64 runs show
    m64a, m64b = r64a.filter_means[1:, 0].double().cpu(), r64b.filter_means[1:, 0].double().cpu()
    se = (r64a.filter_variance[1:, 0].double().cpu() / n).sqrt()
    rms = ((m64a - m64b) ** 2).mean().sqrt() / math.sqrt(2.0)
    sigma = torch.maximum(se, rms.expand_as(se))
    for other in (r32, runs["f32b"]):
        d = (other.filter_means[1:, 0].double().cpu() - m64a).abs()
        assert (d <= 8.0 * math.sqrt(2.0) * sigma + 1e-5 * m64a.abs() + 1e-6).all(), (d / sigma).max()
    ll64 = [r64a.loglikelihood.item(), r64b.loglikelihood.item()]
    spread = abs(ll64[0] - ll64[1])
    for other in (r32, runs["f32b"]):
        assert abs(other.loglikelihood.item() - ll64[0]) <= 8.0 * spread + 0.1, (other.loglikelihood.item(), ll64)
    # the oracle's float64 log-likelihood of the same data (CPU, tools/bench_reference_ll.py) - what bench.py checks itself against
    exp = bench.EXPECTED_LL["apf_lgo_1m"]
    for v in ll64 + [r32.loglikelihood.item(), runs["f32b"].loglikelihood.item()]:
        assert abs(v - exp["loglikelihood"]) <= exp["tol"], (v, exp)


def test_config5_full_length_against_the_exact_kalman_filter_of_every_theta():
    from pyfilter_amd import timeseries as ts
    from pyfilter_amd.filters.particle import APF, proposals
    from pyfilter_amd.timeseries import models

    b, n, t_len = 1024, 8192, 500
    gen = torch.Generator().manual_seed(8)
    kappa = 0.01 + 0.05 * torch.rand(b, generator=gen, dtype=F64)
    gamma = 0.05 * torch.randn(b, generator=gen, dtype=F64)
    sigma = 0.03 + 0.04 * torch.rand(b, generator=gen, dtype=F64)
    x, ys = 0.0, []
    for _ in range(t_len):
        x = x * math.exp(-0.025) + 0.05 * math.sqrt((1 - math.exp(-0.05)) / 0.05) * torch.randn((), generator=gen).item()
        ys.append(x + 0.05 * torch.randn((), generator=gen).item())
    y = torch.tensor(ys, dtype=F64)
    t = lambda v: v.to(device="cuda", dtype=F32)  # noqa: E731
    ssm = ts.LinearStateSpaceModel(models.OrnsteinUhlenbeck(t(kappa), t(gamma), t(sigma), dt=1.0),
                                   (torch.tensor(1.0, device="cuda"), torch.tensor(0.05, device="cuda")))
    filt = APF(ssm, n, proposal=proposals.LinearGaussianObservations(), seed=5)
    filt.set_batch_shape(torch.Size([b]))
    res = filt.batch_filter(y.to(device="cuda", dtype=F32), bar=False)
    torch.cuda.synchronize()
    tr = _trace(t_len)
    assert all(r["MULTI"] == 1 and r["tbytes"] == 4 for r in tr)
    assert all(r["SPEC"] == 1 for r in tr[:-1]), "the APF observed -> observed production kernel did not carry the run"
    e = torch.exp(-kappa)
    q = sigma ** 2 * (1.0 - torch.exp(-2.0 * kappa)) / (2.0 * kappa)
    m, p = gamma.clone(), sigma ** 2 / (2.0 * kappa)
    ll, means, sds = torch.zeros(b, dtype=F64), [], []
    for k in range(t_len):
        m, p = gamma + (m - gamma) * e, p * e * e + q
        s = p + 0.05 ** 2
        ll += -0.5 * (math.log(2.0 * math.pi) + s.log() + (y[k] - m) ** 2 / s)
        gain = p / s
        m, p = m + gain * (y[k] - m), (1.0 - gain) * p
        means.append(m.clone())
        sds.append(p.sqrt())
    got_ll, got_m = res.loglikelihood.cpu().double(), res.filter_means[1:, :, 0].cpu().double()
    assert torch.isfinite(got_ll).all() and torch.isfinite(got_m).all()
    err = (got_ll - ll).abs()
    grow = math.sqrt(t_len / 60.0)  # (the T = 60 bars of test_config5_theta_shard_full_size_..., Monte-Carlo error ~ sqrt(T))
    print("config 5, T = 500: |ll - Kalman| max", err.max().item(), "mean", err.mean().item(), "ll range", ll.min().item(), ll.max().item())
    assert (err <= 0.25 * grow + 2e-3 * ll.abs()).all() and err.mean().item() < 0.08 * grow, (err.max().item(), err.mean().item())
    # unbiasedness of exp(ll) over the 1 024 independent filters (each against ITS exact value)
    r = (got_ll - ll).exp()
    assert abs(r.mean().item() - 1.0) < 5.0 * r.std().item() / math.sqrt(b) + 0.01, (r.mean().item(), r.std().item())
    dm = (got_m - torch.stack(means)).abs()
    assert (dm <= 24.0 * torch.stack(sds) / math.sqrt(n) + 1e-4).all(), (dm / (torch.stack(sds) / math.sqrt(n))).max().item()


def test_config3_full_length_invariants_and_the_float64_run_of_eight_series():
    from pyfilter_amd.filters.particle import APF, proposals

    n, b, t_len = 65536, 64, 1000
    case = dict(name="cfg3", model="sv_batched", filter="apf", proposal="bootstrap", N=n, B=b, T=t_len, ess_threshold=0.9, seed=303,
                param_step_scale=0.05)
    y = simulate(case, build_spec(case, F64))
    assert y.shape == (t_len, b)

    def run(dtype, nb, seed):
        c = dict(case, B=nb)
        f = APF(build_ssm_from_case(c, dtype, "cuda"), n, proposal=proposals.Bootstrap(), seed=seed)
        f.set_batch_shape(torch.Size([nb]))
        r = f.batch_filter(y[:, :nb].to(dtype).cuda(), bar=False)
        torch.cuda.synchronize()
        return r

    r32 = run(F32, b, 1)
    tr = _trace(t_len)
    assert all(r["tbytes"] == 4 and r["D"] == 1 and r["MK"] == 1 and r["MULTI"] == 1 for r in tr)
    assert [r["SPEC"] for r in tr] == [1] * (len(tr) - 1) + [0]
    r64a, r64b = run(F64, 8, 3), run(F64, 8, 4)
    for r in (r32, r64a, r64b):
        assert torch.isfinite(r.filter_means).all() and torch.isfinite(r.loglikelihood).all() and (r.filter_variance >= 0).all()
        idx = r.latest_state.previous_indices
        assert _sorted_in_range(idx, n)
        xs, mm = r.latest_state.timeseries_state.value, r.filter_means[-1, :, 0]
        assert ((xs.min(dim=0)[0] <= mm) & (mm <= xs.max(dim=0)[0])).all() and (xs > 0).all()
    m64a, m64b = r64a.filter_means[1:, :, 0].double().cpu(), r64b.filter_means[1:, :, 0].double().cpu()
    se = (r64a.filter_variance[1:, :, 0].double().cpu() / n).sqrt()
    rms = ((m64a - m64b) ** 2).mean(0).sqrt() / math.sqrt(2.0)
    sigma = torch.maximum(se, rms.expand_as(se))
    d = (r32.filter_means[1:, :8, 0].double().cpu() - m64a).abs()
    assert (d <= 8.0 * math.sqrt(2.0) * sigma + 1e-5 * m64a.abs()).all(), (d / sigma).max()
    lla, llb = r64a.loglikelihood.double().cpu(), r64b.loglikelihood.double().cpu()
    spread = (lla - llb).abs().max().item()
    assert ((r32.loglikelihood[:8].double().cpu() - lla).abs() <= 8.0 * spread + 0.1 + 1e-4 * lla.abs()).all(), \
        ((r32.loglikelihood[:8].double().cpu() - lla).abs().max().item(), spread)


def test_config4_two_hundred_steps_of_lorenz_at_4m_particles_multinomial():
    from pyfilter_amd import resampling
    from pyfilter_amd.filters.particle import SISR, proposals

    n, t_len = 1 << 22, 200
    case = dict(name="cfg4", model="lorenz", filter="sisr", proposal="bootstrap", N=n, B=1, T=t_len, ess_threshold=0.9, seed=404)
    y = simulate(case, build_spec(case, F64))

    def run(dtype, resampler, seed):
        ssm = build_ssm_from_case(dict(model="lorenz", B=1), dtype, "cuda")
        f = SISR(ssm, n, proposal=proposals.Bootstrap(), resampling=resampler, ess_threshold=0.9, seed=seed)
        r = f.batch_filter(y.to(dtype).cuda(), bar=False)
        torch.cuda.synchronize()
        return r

    r32m = run(F32, resampling.multinomial, 11)
    tr = _trace(t_len)
    assert all(r["MODE"] == 1 and r["D"] == 3 and r["tbytes"] == 4 and r["SPEC"] == 2 for r in tr)
    r64a, r64b = run(F64, resampling.multinomial, 21), run(F64, resampling.multinomial, 22)
    r32s = run(F32, resampling.systematic, 12)
    for r in (r32m, r64a, r64b, r32s):
        assert torch.isfinite(r.filter_means).all() and torch.isfinite(r.loglikelihood).all() and (r.filter_variance >= 0).all()
        assert _sorted_in_range(r.latest_state.previous_indices, n)
    m64a, m64b = r64a.filter_means[1:].double().cpu(), r64b.filter_means[1:].double().cpu()
    sigma = ((m64a - m64b) ** 2).mean(dim=0).sqrt() / math.sqrt(2.0) + (r64a.filter_variance[1:].double().cpu() / n).sqrt().mean(dim=0)
    for other in (r32m, r32s):
        d = (other.filter_means[1:].double().cpu() - m64a).abs()
        assert (d <= 8.0 * math.sqrt(2.0) * sigma + 1e-5 * m64a.abs()).all(), (d / sigma).max()
    spread = abs(r64a.loglikelihood.item() - r64b.loglikelihood.item())
    for other in (r32m, r32s):
        assert abs(other.loglikelihood.item() - r64a.loglikelihood.item()) < 8.0 * spread + 0.01 * t_len, \
            (other.loglikelihood.item(), r64a.loglikelihood.item(), spread)
    # the filter tracks the observed components: A x = 0.8 (x_1, x_3) against y within a few observation standard deviations
    pred = 0.8 * r32m.filter_means[1:][:, [0, 2]].double().cpu()
    assert ((pred - y).abs() <= 6.0 * math.sqrt(0.1)).all()
    _ = cpu_ref  # (the oracle is not run at this size: the step-level checks at 2^22 are tests/test_production_kernels_gpu.py's)
