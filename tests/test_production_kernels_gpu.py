"""Identical-draw parity of the *production* float32 step-kernel instantiations - the ones ``bench.py`` times.

The specialised kernels (``SPEC = 1``: APF observed -> observed, ``SPEC = 2``: SISR observed; closed-form ``MK`` foldings)
are selected only for float32 runs that draw their normals from Philox, so the tape-driven parity tests never execute
them.  Here the run keeps its Philox normals (only the resampling uniforms are injected, which the selection ignores),
``pf_debug_draw_normals`` dumps exactly the normals the kernel consumed, and the oracle (``oracle/cpu_ref.py``,
``pyfilter/filters/particle/apf.py:25-46``, ``sisr.py:14-56``) is teacher-forced with them from the same state:

* APF: a 2-step ``batch_filter(y[t:t+2], init_state=...)`` - step 0 takes ``SPEC = 1`` (its output is the state left in
  the run's other buffer), the last step the generic variant;
* SISR: a 1-step run (``SPEC = 2``).

``pf_debug_launch_trace`` asserts which instantiation really ran.  Bars (those of ``test_float32_teacher_forced_steps``):
x within 1e-5 of the state's scale and w within 2e-5 rel + 2e-4 where the ancestors agree, ancestor flips <= 2e-4."""
import math

import pytest
import torch

from oracle import cpu_ref
from oracle.cases import FUSED_CASES as CASES, build_spec  # (lg1d_o2_*: the torch route, tests/test_torch_route_golden.py)
from pyfilter_amd import ops
from tests.helpers import DT, build_filter_from_case, build_ssm_from_case, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _per_step_route(monkeypatch):
    """These tests pin the STEP kernel's production instantiations (asserted through the launch trace), also at the golden
    cases' small particle counts - which the library would otherwise hand to the column-persistent kernel
    (tests/test_column_route_gpu.py covers that one)."""
    from pyfilter_amd.hints import HINTS

    monkeypatch.setattr(HINTS, "route", 1)
F32 = torch.float32


def _teacher_state(es, t, x, w, ll, idx):
    from pyfilter_amd.filters.particle.state import ParticleFilterCorrection
    from pyfilter_amd.timeseries import TimeseriesState

    return ParticleFilterCorrection(TimeseriesState(t, x, es), w, ll, idx)


def _normals_ref_layout(filt, steps, n, b, d, has_event):
    """The run's normals in the reference layout ``(steps, N, B, [D])`` (float32 values, exactly what the kernel drew)."""
    seed = filt._last_run["seed_eff"]  # base seed + the run's epoch word: what the kernels keyed Philox with
    z = ops.debug_draw_normals(seed, steps, n, b, d, F32, "cuda")  # (steps, D, B, N)
    z = z.permute(0, 3, 2, 1)
    return (z if has_event else z[..., 0]).cpu()


def _oracle_step(spec, case, y_t, x, w, prev_idx, z, u, dtype):
    """One teacher-forced oracle step from float32 inputs, evaluated in ``dtype``."""
    x, w, z, u, y_t = x.to(dtype), w.clone().to(dtype), z.to(dtype), u.to(dtype), y_t.to(dtype)
    if bool(y_t.isnan().all()):
        idx = torch.arange(x.shape[0]).unsqueeze(-1).expand(w.shape)  # APF.predict hands out identity ancestors (apf.py:18-23)
        if case["filter"] == "sisr":
            x, w, _, idx, _ = cpu_ref.sisr_predict(spec, x, w, prev_idx, u, case["ess_threshold"] * x.shape[0])
        xn, wn, ll = cpu_ref.propagate_only_step(spec, x, w, z)
        return xn, wn, ll, idx
    if case["filter"] == "apf":
        return cpu_ref.apf_step(spec, case["proposal"], y_t, x, w, z, u)
    xn, wn, ll, idx, _ = cpu_ref.sisr_step(spec, case["proposal"], y_t, x, w, prev_idx, z, u, case["ess_threshold"] * x.shape[0])
    return xn, wn, ll, idx


def _compare_step(tag, x_gpu, w_gpu, ll_gpu, idx_gpu, ref64, ref32, n):
    """One teacher-forced step of the kernel against the oracle evaluated in float64 and in float32 (the reference's own
    arithmetic) on the same inputs and draws.  Returns ``(flips, allowance)``.

    A particle *matches* when its ancestor equals the float64 oracle's - or the float32 oracle's: a grid position within
    an ulp of a CDF boundary may legitimately fall either way - and its new state / log-weight are within the bars of
    the oracle that owns that ancestor (x: 1e-5 of the state's scale, w: 2e-5 relative + 2e-4).  With the ancestors at
    hand a matching ancestor with an off value fails at once (the runs record their states, so the intermediate step of a
    2-step APF run - the SPEC = 1 kernel - hands out its ancestors too; without ancestors a non-matching particle would
    count as a flip).  Allowance: 2e-4 of the particles (the bar of the golden
    teacher-forced test, N <= 1000) or 3 x the flips between the two oracles themselves - at 2^20 particles a CDF
    increment is 16 float32 ulps, so the reference's own float32 path flips that often against exact arithmetic."""
    x64, w64, ll64, idx64 = ref64
    x32, w32, _, idx32 = ref32
    scale = x64.abs().max().item()
    xg, wg = x_gpu.double(), w_gpu.double()

    def close(xr, wr):
        dx = (xg - xr.double()).abs() <= 1e-5 * scale + 1e-6
        if dx.dim() > wr.dim():
            dx = dx.all(dim=-1)
        wr = wr.double()
        dw = ((wg - wr).abs() <= 2e-5 * wr.abs() + 2e-4) | ~torch.isfinite(wr)
        return dx, dw

    dx64, dw64 = close(x64, w64)
    dx32, dw32 = close(x32, w32)
    if idx_gpu is not None:
        on64 = idx_gpu == idx64
        on32 = ~on64 & (idx_gpu == idx32)
        assert (dx64 | ~on64).all() and (dx32 | ~on32).all(), f"{tag}: x differs on identical ancestors"
        assert (dw64 | ~on64).all() and (dw32 | ~on32).all(), f"{tag}: w differs on identical ancestors"
        match = on64 | on32
    else:
        match = (dx64 & dw64) | (dx32 & dw32)
    flips = int((~match).sum())
    oracle_flips = int((idx64 != idx32).sum())
    b = max(1, ll64.numel())
    allowance = max(2, int(2e-4 * n * b), 3 * oracle_flips)
    assert flips <= allowance, f"{tag}: {flips} ancestor flips (allowance {allowance}, the two oracles differ in {oracle_flips})"
    if ll_gpu is not None:
        # every particle that follows another ancestor than the float64 oracle's moves the estimate by O(1 / N)
        moved = flips + oracle_flips
        torch.testing.assert_close(ll_gpu.double(), ll64, rtol=1e-4, atol=1e-4 + 10.0 * moved / n)
    return flips, allowance


def _expect_variant(case, steps_observed, first):
    """(SPEC of the first launch)"""
    if not steps_observed[0]:
        return 0
    if case["filter"] == "apf":
        return 1 if (len(steps_observed) > 1 and steps_observed[1]) else 0
    return 2


# (cases observed at every move: a thinned run's propagate-only moves are the NaN-observation launches the *_nan cases cover)
@pytest.mark.parametrize("name", [c["name"] for c in CASES if c.get("observe_every_step", 1) == 1])
def test_production_step_kernels_match_oracle_on_their_own_draws(name):
    case = next(c for c in CASES if c["name"] == name)
    dt = "f32" if "f32" in case["dtypes"] else "f64"  # the teacher states: the reference's own (cast to float32 if need be)
    g = load_golden(name, dt)
    spec64, spec32 = build_spec(case, torch.float64), build_spec(case, F32)
    n, b = case["N"], case["B"]
    y = g["y"].to(F32)
    t_len = y.shape[0]
    apf = case["filter"] == "apf"
    run_len = 2 if apf else 1
    d = max(1, spec64.dim)
    has_event = spec64.dim > 0
    seen_spec = set()
    # recorded states: the kernels keep every state of the run WITH the ancestors that led to it (pf_filter_args.ring), so
    # the intermediate step of the 2-step APF runs - the SPEC = 1 kernel - hands out its ancestors for a direct comparison
    filt = build_filter_from_case(case, g, F32, "cuda", tape=False, record_states=True)
    filt.set_tape(u=g["u_tape"].to(F32))  # uniforms injected, normals stay Philox: the production kernels are selected
    es = filt._model.hidden.event_shape
    for t in range(0, t_len - run_len + 1):
        if t == 0:
            x_prev, w_prev = g["x0"].to(F32), torch.zeros(g["x0"].shape[:2] if has_event else g["x0"].shape, dtype=F32)
            idx_prev = torch.arange(n).unsqueeze(-1).expand(n, b).contiguous()
            ll_prev = torch.zeros(b, dtype=F32)
        else:
            x_prev, w_prev = g["step_x"][t - 1].to(F32), g["step_w"][t - 1].to(F32)
            idx_prev, ll_prev = g["step_idx"][t - 1], g["step_ll"][t - 1].to(F32)
        w_prev = torch.nan_to_num(w_prev, nan=-math.inf, posinf=-math.inf)
        prev = _teacher_state(es, t, x_prev.cuda(), w_prev.clone().cuda(), ll_prev.cuda(), idx_prev.cuda())
        res = filt.batch_filter(y[t:t + run_len].cuda(), bar=False, init_state=prev)
        torch.cuda.synchronize()
        trace = ops.debug_launch_trace(run_len)
        observed = [not bool(y[t + s].isnan().all()) for s in range(run_len)]
        assert [r["step"] for r in trace] == list(range(run_len)) and all(r["tbytes"] == 4 for r in trace)
        assert trace[0]["SPEC"] == _expect_variant(case, observed, True), (trace, observed)
        seen_spec.add(trace[0]["SPEC"])
        z = _normals_ref_layout(filt, run_len, n, b, d, has_event)
        u = g["u_tape"].to(F32)
        plan = filt._last_run["plan"]
        last = res.latest_state
        if apf:
            # step 0 (SPEC = 1 when both steps are observed): its recorded state, ancestors included
            mid = res.states[-2]
            x1, w1 = mid.timeseries_state.value.cpu(), mid.weights.cpu()
            r64 = _oracle_step(spec64, case, y[t], x_prev, w_prev, idx_prev, z[0], u[t], torch.float64)
            r32 = _oracle_step(spec32, case, y[t], x_prev, w_prev, idx_prev, z[0], u[t], F32)
            _compare_step(f"{name} t={t} step0", x1, w1, plan.ll_steps[0].cpu(), mid.previous_indices.cpu(), r64, r32, n)
            # step 1 (generic variant: the run's last step), teacher-forced from the kernel's own step-0 state
            r64 = _oracle_step(spec64, case, y[t + 1], x1, w1, r64[3], z[1], u[t + 1], torch.float64)
            r32 = _oracle_step(spec32, case, y[t + 1], x1, w1, r32[3], z[1], u[t + 1], F32)
            _compare_step(f"{name} t={t} step1", last.timeseries_state.value.cpu(), last.weights.cpu(),
                          last.get_loglikelihood().cpu(), last.previous_indices.cpu(), r64, r32, n)
        else:
            r64 = _oracle_step(spec64, case, y[t], x_prev, w_prev, idx_prev, z[0], u[t], torch.float64)
            r32 = _oracle_step(spec32, case, y[t], x_prev, w_prev, idx_prev, z[0], u[t], F32)
            _compare_step(f"{name} t={t}", last.timeseries_state.value.cpu(), last.weights.cpu(),
                          last.get_loglikelihood().cpu(), last.previous_indices.cpu(), r64, r32, n)
    assert (1 if apf else 2) in seen_spec, f"the specialised kernel never ran: {seen_spec}"


def _bench_shape_case(model, filt_name, prop, n, b, ess=0.9):
    return dict(name=f"{model}_{filt_name}_{prop}", model=model, filter=filt_name, proposal=prop, N=n, B=b, T=4,
                ess_threshold=ess, seed=4242, dtypes=("f32",))


@pytest.mark.parametrize("model,filt_name,prop,n,b", [
    ("sine", "apf", "lgo", 1 << 20, 1),          # BASELINE config 2: the headline instantiation (FAST, MK = 2, SPEC = 1)
    ("sv_batched", "apf", "bootstrap", 65536, 8),  # config 3 (8 of the 64 series): generic MK = 1, SPEC = 1
    ("lorenz", "sisr", "bootstrap", 1 << 20, 1),  # config 4's model: D = 3, SPEC = 2
    ("ou_batched", "sisr", "lgo", 8192, 16),      # config 5's theta-shard shape (16 of 128 columns): SPEC = 2
    ("lg1d", "apf", "bootstrap", 1 << 18, 2),     # affine closed form (MK = 1) with the bootstrap proposal
])
def test_production_step_kernels_at_benchmark_shapes(model, filt_name, prop, n, b):
    """The same check on multi-tile shapes, from the filter's own Philox-initialised state and then from a state with
    non-trivial weights (the second run starts where the first ended)."""
    from oracle.cases import simulate

    case = _bench_shape_case(model, filt_name, prop, n, b)
    spec64, spec32 = build_spec(case, torch.float64), build_spec(case, F32)
    y = simulate(case, spec64).to(F32)
    apf = filt_name == "apf"
    run_len = 2 if apf else 1
    d, has_event = max(1, spec64.dim), spec64.dim > 0
    gen = torch.Generator().manual_seed(77)
    u = torch.rand((8, b), generator=gen, dtype=F32)
    ssm = build_ssm_from_case(case, F32, "cuda")
    from pyfilter_amd.filters.particle import APF, SISR, proposals

    cls = APF if apf else SISR
    filt = cls(ssm, n, proposal={"bootstrap": proposals.Bootstrap, "lgo": proposals.LinearGaussianObservations}[prop](),
               ess_threshold=case["ess_threshold"], seed=99, record_states=True)  # (recorded: the intermediate ancestors)
    filt.set_batch_shape(torch.Size([b]))
    filt.set_tape(u=u)
    state = filt.initialize()
    t = 0
    for _ in range(2):
        xs = state.timeseries_state.value.cpu()
        ws = state.weights.cpu().clone()
        idx_prev = state.previous_indices.cpu()
        res = filt.batch_filter(y[t:t + run_len].cuda(), bar=False, init_state=state)
        torch.cuda.synchronize()
        trace = ops.debug_launch_trace(run_len)
        assert trace[0]["SPEC"] == (1 if apf else 2) and trace[0]["tbytes"] == 4 and trace[0]["D"] == d, trace
        z = _normals_ref_layout(filt, run_len, n, b, d, has_event)
        plan = filt._last_run["plan"]
        last = res.latest_state
        if apf:
            mid = res.states[-2]  # the SPEC = 1 step's own output, ancestors included
            x1, w1 = mid.timeseries_state.value.cpu(), mid.weights.cpu()
            r64 = _oracle_step(spec64, case, y[t], xs, ws, idx_prev, z[0], u[t], torch.float64)
            r32 = _oracle_step(spec32, case, y[t], xs, ws, idx_prev, z[0], u[t], F32)
            _compare_step(f"{case['name']} t={t} step0", x1, w1, plan.ll_steps[0].cpu(), mid.previous_indices.cpu(), r64, r32, n)
            r64 = _oracle_step(spec64, case, y[t + 1], x1, w1, r64[3], z[1], u[t + 1], torch.float64)
            r32 = _oracle_step(spec32, case, y[t + 1], x1, w1, r32[3], z[1], u[t + 1], F32)
        else:
            r64 = _oracle_step(spec64, case, y[t], xs, ws, idx_prev, z[0], u[t], torch.float64)
            r32 = _oracle_step(spec32, case, y[t], xs, ws, idx_prev, z[0], u[t], F32)
        _compare_step(f"{case['name']} t={t} last", last.timeseries_state.value.cpu(), last.weights.cpu(),
                      last.get_loglikelihood().cpu(), last.previous_indices.cpu(), r64, r32, n)
        state = last
        t += run_len


# ---- BASELINE configs 3 and 4 as written -------------------------------------------------------------------------------
def _lorenz_ssm(dtype):
    case = dict(model="lorenz", B=1)
    return build_ssm_from_case(case, dtype, "cuda")


def test_config4_lorenz_sisr_multinomial_4m_particles():
    """BASELINE configs[3] as written (short T): Lorenz-63 Euler-Maruyama, SISR + Bootstrap, **multinomial**, 2^22
    particles, float32 - the ``D = 3, MODE = 1`` step kernel.  ``torch.multinomial`` is not injectable
    (``pyfilter/resampling.py:55-65``), so: (i) the kernel's own step is teacher-forced through the oracle *given its
    ancestors* and its own Philox normals (propagation + weights of every particle, identical draws); (ii) the
    ancestors are a valid multinomial draw - sorted, in range, offspring counts consistent with N W (chi-square over
    weight-deciles); (iii) means / log-likelihood agree with the float64 run of the same filter and with the systematic
    run within Monte-Carlo error."""
    from oracle.cases import simulate
    from pyfilter_amd import resampling
    from pyfilter_amd.filters.particle import SISR, proposals

    n, t_len = 1 << 22, 6
    case = dict(name="cfg4", model="lorenz", filter="sisr", proposal="bootstrap", N=n, B=1, T=t_len, ess_threshold=0.9, seed=404)
    spec64 = build_spec(case, torch.float64)
    y = simulate(case, spec64)

    def run(dtype, resampler, seed=11, init_state=None, steps=slice(None)):
        f = SISR(_lorenz_ssm(dtype), n, proposal=proposals.Bootstrap(), resampling=resampler, ess_threshold=0.9, seed=seed)
        r = f.batch_filter(y[steps].to(dtype).cuda(), bar=False, init_state=init_state)
        torch.cuda.synchronize()
        return f, r

    f32m, r32m = run(F32, resampling.multinomial)
    tr = ops.debug_launch_trace(t_len)
    assert all(r["MODE"] == 1 and r["D"] == 3 and r["tbytes"] == 4 and r["SPEC"] == 2 for r in tr), tr
    _, r64a = run(torch.float64, resampling.multinomial, seed=21)
    _, r64b = run(torch.float64, resampling.multinomial, seed=22)
    _, r32s = run(F32, resampling.systematic)

    for r in (r32m, r64a, r64b, r32s):
        assert torch.isfinite(r.filter_means).all() and torch.isfinite(r.loglikelihood).all()
    # Monte-Carlo scale: the spread of two independent float64 runs (the y component is unobserved: its error is far above
    # the current cloud's sqrt(var / N)), per component, rms over time
    m64a, m64b = r64a.filter_means[1:].double().cpu(), r64b.filter_means[1:].double().cpu()
    sigma = ((m64a - m64b) ** 2).mean(dim=0).sqrt() / math.sqrt(2.0) + (r64a.filter_variance[1:].double().cpu() / n).sqrt().mean(dim=0)
    for other in (r32m, r32s):
        d = (other.filter_means[1:].double().cpu() - m64a).abs()
        assert (d <= 8.0 * math.sqrt(2.0) * sigma + 1e-5 * m64a.abs()).all(), (d / sigma).max()
    ll_spread = abs(r64a.loglikelihood.item() - r64b.loglikelihood.item())
    for other in (r32m, r32s):
        assert abs(other.loglikelihood.item() - r64a.loglikelihood.item()) < 8.0 * ll_spread + 0.01 * t_len

    # (i) + (ii): one more multinomial step from the final state, teacher-forced given the kernel's ancestors
    state = r32m.latest_state
    xs, ws = state.timeseries_state.value.cpu(), state.weights.cpu().clone()
    y_next = simulate(dict(case, T=t_len + 1), spec64)[t_len:t_len + 1].to(F32)
    f2 = SISR(_lorenz_ssm(F32), n, proposal=proposals.Bootstrap(), resampling=resampling.multinomial, ess_threshold=1.1, seed=12)
    r2 = f2.batch_filter(y_next.cuda(), bar=False, init_state=state)
    torch.cuda.synchronize()
    tr = ops.debug_launch_trace(1)
    assert tr[0]["MODE"] == 1 and tr[0]["D"] == 3 and tr[0]["SPEC"] == 2, tr
    last = r2.latest_state
    anc = last.previous_indices.cpu()
    assert (anc[1:] >= anc[:-1]).all() and anc.min() >= 0 and anc.max() <= n - 1
    z = _normals_ref_layout(f2, 1, n, 1, 3, True)[0, :, 0]  # (N, 3): the filter is unbatched
    x_r = xs.double()[anc]
    x_new, wi = cpu_ref.bootstrap_sample_and_weight(spec64, y_next[0].double(), x_r, z.double())
    scale = x_new.abs().max().item()
    assert ((last.timeseries_state.value.cpu().double() - x_new).abs() <= 1e-5 * scale + 1e-6).all()
    fin = torch.isfinite(wi)
    assert ((last.weights.cpu().double() - wi).abs()[fin] <= 2e-5 * wi[fin].abs() + 2e-4).all()
    # offspring counts vs N W, pooled over 16 weight-quantile groups: chi-square (groups expecting < 50 offspring dropped)
    W = cpu_ref.normalize(ws.double())
    counts = torch.bincount(anc, minlength=n).double()
    groups = W.argsort().reshape(16, -1)
    exp_g = (W[groups] * n).sum(dim=1)
    obs_g = counts[groups].sum(dim=1)
    keep = exp_g >= 50.0
    dof = int(keep.sum()) - 1
    chi2 = ((obs_g - exp_g)[keep] ** 2 / exp_g[keep]).sum().item()
    assert dof >= 3 and chi2 < dof + 8.0 * math.sqrt(2.0 * dof), (chi2, dof)


def test_config3_sv_64_series_full_size_invariants():
    """BASELINE configs[2] at full size (64 independent series x 65 536 particles, APF + Bootstrap, float32, Philox):
    the production instantiation ran (generic MK = 1, SPEC = 1), outputs finite, ancestors sorted / in range, every
    series' mean inside its particle cloud, the same seed reproduces bit for bit, two seeds agree within Monte-Carlo
    error, and the float64 run of the same filter agrees within Monte-Carlo error."""
    from oracle.cases import simulate
    from pyfilter_amd.filters.particle import APF, proposals

    n, b, t_len = 65536, 64, 12
    case = dict(name="cfg3", model="sv_batched", filter="apf", proposal="bootstrap", N=n, B=b, T=t_len, ess_threshold=0.9, seed=303,
                param_step_scale=0.05)  # 64 distinct, moderate (kappa, gamma, sigma, mu) rows
    y = simulate(case, build_spec(case, torch.float64))

    def run(dtype, seed):
        f = APF(build_ssm_from_case(case, dtype, "cuda"), n, proposal=proposals.Bootstrap(), seed=seed)
        f.set_batch_shape(torch.Size([b]))
        r = f.batch_filter(y.to(dtype).cuda(), bar=False)
        torch.cuda.synchronize()
        return r

    r1 = run(F32, 1)
    tr = ops.debug_launch_trace(t_len)
    assert all(r["tbytes"] == 4 and r["D"] == 1 and r["FAST"] == 0 and r["MK"] == 1 for r in tr), tr
    assert [r["SPEC"] for r in tr] == [1] * (t_len - 1) + [0], tr
    r2, r1b, r64 = run(F32, 2), run(F32, 1), run(torch.float64, 3)
    for r in (r1, r2, r64):
        assert torch.isfinite(r.filter_means).all() and torch.isfinite(r.loglikelihood).all()
        idx = r.latest_state.previous_indices
        assert (idx[1:] >= idx[:-1]).all() and idx.min() >= 0 and idx.max() <= n - 1
        xs = r.latest_state.timeseries_state.value
        m = r.filter_means[-1, :, 0]
        assert ((xs.min(dim=0)[0] <= m) & (m <= xs.max(dim=0)[0])).all()
    assert torch.equal(r1.filter_means, r1b.filter_means) and torch.equal(r1.loglikelihood, r1b.loglikelihood)
    se = (r64.filter_variance[1:].double() / n).sqrt()
    for other in (r2, r64):
        d = (r1.filter_means[1:].double() - other.filter_means[1:].double()).abs()
        assert (d <= 8.0 * se + 1e-5).all(), (d / (se + 1e-12)).max()
        assert ((r1.loglikelihood.double() - other.loglikelihood.double()).abs() < 0.05).all()


def test_config5_theta_shard_full_size_against_the_exact_kalman_likelihood():
    """BASELINE configs[4]'s filtering pass as written - 1 024 theta-particles x 8 192 state particles, the OU model of
    tests/inference/models.py, APF + LinearGaussianObservations, float32, Philox draws - against the EXACT answer: the
    model is linear-Gaussian, so every theta-particle's log-likelihood and filter means are a Kalman filter's.  1 024
    independent checks of the production multi-round kernel at the shape the SMC^2 line is quoted on."""
    from pyfilter_amd import timeseries as ts
    from pyfilter_amd.filters.particle import APF, proposals
    from pyfilter_amd.timeseries import models

    b, n, t_len = 1024, 8192, 60
    gen = torch.Generator().manual_seed(8)
    kappa = 0.01 + 0.05 * torch.rand(b, generator=gen, dtype=torch.float64)  # (theta in the posterior's neighbourhood: a
    gamma = 0.05 * torch.randn(b, generator=gen, dtype=torch.float64)         # theta-particle far from the data has a
    sigma = 0.03 + 0.04 * torch.rand(b, generator=gen, dtype=torch.float64)   # degenerate filter and O(1) Monte-Carlo error)
    x, ys = 0.0, []
    for _ in range(t_len):
        x = x * math.exp(-0.025) + 0.05 * math.sqrt((1 - math.exp(-0.05)) / 0.05) * torch.randn((), generator=gen).item()
        ys.append(x + 0.05 * torch.randn((), generator=gen).item())
    y = torch.tensor(ys, dtype=torch.float64)
    t = lambda v: v.to(device="cuda", dtype=F32)  # noqa: E731
    ssm = ts.LinearStateSpaceModel(models.OrnsteinUhlenbeck(t(kappa), t(gamma), t(sigma), dt=1.0),
                                   (torch.tensor(1.0, device="cuda"), torch.tensor(0.05, device="cuda")))
    filt = APF(ssm, n, proposal=proposals.LinearGaussianObservations(), seed=5)
    filt.set_batch_shape(torch.Size([b]))
    res = filt.batch_filter(y.to(device="cuda", dtype=F32), bar=False)
    torch.cuda.synchronize()
    tr = ops.debug_launch_trace(4)[-1]
    assert tr["MULTI"] == 1 and tr["tbytes"] == 4 and tr["SPEC"] in (0, 1), tr  # the multi-round production kernel ran
    # exact: x' = gamma + (x - gamma) e^{-kappa} + s_d e, s_d^2 = sigma^2 (1 - e^{-2 kappa}) / (2 kappa); x0 ~ N(gamma, sigma^2 / (2 kappa))
    e = torch.exp(-kappa)
    q = sigma ** 2 * (1.0 - torch.exp(-2.0 * kappa)) / (2.0 * kappa)
    m, p = gamma.clone(), sigma ** 2 / (2.0 * kappa)
    ll, means = torch.zeros(b, dtype=torch.float64), []
    for k in range(t_len):
        m, p = gamma + (m - gamma) * e, p * e * e + q
        s = p + 0.05 ** 2
        ll += -0.5 * (math.log(2.0 * math.pi) + s.log() + (y[k] - m) ** 2 / s)
        gain = p / s
        m, p = m + gain * (y[k] - m), (1.0 - gain) * p
        means.append(m.clone())
    got_ll = res.loglikelihood.cpu().double()
    got_m = res.filter_means[1:, :, 0].cpu().double()
    assert torch.isfinite(got_ll).all() and torch.isfinite(got_m).all()
    # Monte-Carlo error of a particle filter's log-likelihood at 8 192 particles with the optimal proposal: a few 1e-2
    err = (got_ll - ll).abs()
    print("config 5 full size: |ll - Kalman| max", err.max().item(), "mean", err.mean().item(), "ll range", ll.min().item(), ll.max().item())
    assert (err <= 0.25 + 2e-3 * ll.abs()).all() and err.mean().item() < 0.08, (err.max().item(), err.mean().item())
    dm = (got_m - torch.stack(means)).abs()
    post_sd = torch.stack([p.sqrt()] * t_len)
    assert (dm <= 8.0 * post_sd / math.sqrt(n) * 3.0 + 1e-4).all(), (dm / (post_sd / math.sqrt(n))).max().item()


@pytest.mark.parametrize("route", ["per_step", "auto"])
@pytest.mark.parametrize("skewed", [False, True])
@pytest.mark.parametrize("filt_name,prop", [("sisr", "lgo"), ("apf", "lgo"), ("sisr", "bootstrap"), ("apf", "bootstrap")])
@pytest.mark.parametrize("hid,d,o", [("rw", 2, 0), ("rw", 2, 1), ("rw", 2, 2), ("rw", 2, 3), ("rw", 3, 3), ("lorenz", 3, 0),
                                     ("lorenz", 3, 1), ("lorenz", 3, 2), ("lorenz", 3, 3)])
def test_production_kernels_random_linear_observations(hid, d, o, filt_name, prop, skewed, route, monkeypatch):
    """Every (D, O) the fused kernels accept for a vector state, with a DENSE random observation matrix, offset and noise
    scales (``o = 0``: a scalar observation, ``event_shape = Size([])``) - the float32 production instantiations on their own
    Philox draws against the oracle, from uniform weights (SISR does not resample) and from skewed ones (it does).  Round 5:
    the first dense 3x3 fixture exposed a miscompiled packed multiply-add in the optimal proposal's 3x3 inverse that every
    sparser matrix hides (profiles/r05_slp_pk_fma_miscompile.txt); the fixtures pin a handful of matrices, this sweeps them."""
    from oracle import models as M
    from pyfilter_amd import timeseries as ts
    from pyfilter_amd.filters.particle import APF, SISR, proposals
    from pyfilter_amd.hints import HINTS
    from pyfilter_amd.timeseries import models

    monkeypatch.setattr(HINTS, "route", 1 if route == "per_step" else 0)
    n, b = 768, 2
    gen = torch.Generator().manual_seed(1000 * d + 100 * o + 10 * len(filt_name) + len(prop) + (5 if skewed else 0))
    od = max(o, 1)
    a = torch.randn(od, d, generator=gen, dtype=torch.float64) * 0.6 + (torch.eye(od, d, dtype=torch.float64) if od <= d else 0.0)
    off = 0.3 * torch.randn(od, generator=gen, dtype=torch.float64)
    s = 0.2 + 0.5 * torch.rand(od, generator=gen, dtype=torch.float64)
    if o == 0:
        a, off, s = a[0], off[0], s[0]
    t = lambda v: v.to(F32).cuda()  # noqa: E731
    if hid == "rw":
        sig = 0.3 + 0.5 * torch.rand(d, generator=gen, dtype=torch.float64)
        centre = torch.zeros(d, dtype=torch.float64)
        hidden = models.RandomWalk(t(sig), initial_mean=t(centre), initial_scale=t(sig), dim=d)
        hid_spec = (M.HID_LINEAR, (torch.zeros(d, dtype=torch.float64), torch.ones(d, dtype=torch.float64), sig), d, 1.0, (centre, sig))
    else:
        centre = torch.tensor([-5.9, -5.5, 24.6], dtype=torch.float64)
        hidden = models.Lorenz63(t(torch.tensor(10.0)), t(torch.tensor(28.0)), t(torch.tensor(8.0 / 3.0)), t(torch.tensor(1.0)), dt=0.01,
                                 initial_mean=t(centre), initial_scale=t(torch.full((3,), math.sqrt(10.0))))
        hid_spec = (M.HID_LORENZ63_EM, (10.0, 28.0, 8.0 / 3.0, 1.0), 3, 0.01, (centre, torch.full((3,), math.sqrt(10.0), dtype=torch.float64)))
    ssm = ts.LinearStateSpaceModel(hidden, (t(a), t(off), t(s)), torch.Size([o]) if o else torch.Size([])).to("cuda")
    spec64 = M.ModelSpec(*hid_spec, M.OBS_LINEAR, (a, off, s), o)
    spec32 = M.ModelSpec(hid_spec[0], tuple(p.to(F32) if isinstance(p, torch.Tensor) else p for p in hid_spec[1]), d, hid_spec[3],
                         tuple(p.to(F32) for p in hid_spec[4]), M.OBS_LINEAR, (a.to(F32), off.to(F32), s.to(F32)), o)
    y_oracle = lambda v: v  # noqa: E731
    if o == 0 and filt_name == "apf" and prop == "lgo":
        # the one combination the reference cannot run: its LinearGaussianObservations.pre_weight mixes (N, B, D) and (N, B)
        # tensors for a scalar observation of a vector state (proposals/linear.py:79-81 - it raises; the oracle restates that
        # line).  The product evaluates what the formula means; the oracle side takes the same observation declared as a
        # vector of length one (event_shape = Size([1]): identical arithmetic, golden case lorenz_o1_apf_lgo)
        as_vec = lambda sp: M.ModelSpec(sp.hidden, sp.hidden_params, sp.dim, sp.dt, sp.init, M.OBS_LINEAR,  # noqa: E731
                                        (sp.obs_params[0].unsqueeze(0), sp.obs_params[1].reshape(1), sp.obs_params[2].reshape(1)), 1)
        spec64, spec32 = as_vec(spec64), as_vec(spec32)
        y_oracle = lambda v: v.unsqueeze(-1)  # noqa: E731
    case = dict(filter=filt_name, proposal=prop, ess_threshold=0.5)
    cls = {"sisr": SISR, "apf": APF}[filt_name]
    p = {"bootstrap": proposals.Bootstrap, "lgo": proposals.LinearGaussianObservations}[prop]()
    filt = cls(ssm, n, proposal=p, ess_threshold=0.5, record_states=True)
    filt.set_batch_shape(torch.Size([b]))
    run_len = 2 if filt_name == "apf" else 1
    u = torch.rand(run_len, b, generator=gen).to(F32)
    filt.set_tape(u=u)
    spread = 0.5 if hid == "rw" else 2.0
    x_prev = (centre + spread * torch.randn(n, b, d, generator=gen, dtype=torch.float64)).to(F32)
    w_prev = (2.5 * torch.randn(n, b, generator=gen)).to(F32) if skewed else torch.zeros(n, b, dtype=F32)
    loc = off + (a * centre).sum(-1) if o == 0 else off + a @ centre
    y = (loc + 0.5 * torch.randn((run_len,) + tuple(loc.shape), generator=gen, dtype=torch.float64)).to(F32)
    idx_prev = torch.arange(n).unsqueeze(-1).expand(n, b).contiguous()
    prev = _teacher_state(filt._model.hidden.event_shape, 0, x_prev.cuda(), w_prev.clone().cuda(), torch.zeros(b).cuda(), idx_prev.cuda())
    res = filt.batch_filter(y.cuda(), bar=False, init_state=prev)
    torch.cuda.synchronize()
    z = _normals_ref_layout(filt, run_len, n, b, d, True)
    last = res.latest_state
    tag = f"{hid} D={d} O={o} {filt_name}+{prop} skewed={skewed} {route}"
    if run_len == 2:
        mid = res.states[-2]
        x1, w1 = mid.timeseries_state.value.cpu(), mid.weights.cpu()
        r64 = _oracle_step(spec64, case, y_oracle(y[0]), x_prev, w_prev, idx_prev, z[0], u[0], torch.float64)
        r32 = _oracle_step(spec32, case, y_oracle(y[0]), x_prev, w_prev, idx_prev, z[0], u[0], F32)
        _compare_step(tag + " step0", x1, w1, None, mid.previous_indices.cpu(), r64, r32, n)
        r64 = _oracle_step(spec64, case, y_oracle(y[1]), x1, w1, r64[3], z[1], u[1], torch.float64)
        r32 = _oracle_step(spec32, case, y_oracle(y[1]), x1, w1, r32[3], z[1], u[1], F32)
    else:
        r64 = _oracle_step(spec64, case, y_oracle(y[0]), x_prev, w_prev, idx_prev, z[0], u[0], torch.float64)
        r32 = _oracle_step(spec32, case, y_oracle(y[0]), x_prev, w_prev, idx_prev, z[0], u[0], F32)
    _compare_step(tag, last.timeseries_state.value.cpu(), last.weights.cpu(), last.get_loglikelihood().cpu(),
                  last.previous_indices.cpu(), r64, r32, n)
