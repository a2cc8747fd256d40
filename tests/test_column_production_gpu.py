"""Identical-draw parity of the *production* float32 COLUMN kernels (``pyfilter_amd/csrc/pf_column.hpp``): the instantiations
with the model kind / filter / proposal folded at compile time (``KIND / FILT / PROP``, RAGGED columns, the 1024-thread
bound) that run at the reference's own operating point - 1 000 theta x 250 - 400 particles
(``examples/stochastic-volatility.ipynb:157``) - against the oracle, step by step, on the draws the kernel itself consumed.

Those kernels are selected only for float32 runs that draw their normals from Philox, so the tape-driven golden suites
reach the column route on its run-time kernel only.  Here (the column-route twin of
``tests/test_production_kernels_gpu.py::test_production_step_kernels_match_oracle_on_their_own_draws``):

* ONE multi-step run of the column kernel (Philox normals; only the resampling uniforms injected, which the selection
  ignores) - ``pf_debug_launch_trace`` asserts ``SPEC == 9`` (column route) and ``FAST == 1`` (a specialised instantiation);
* the run's intermediate states - a column run keeps no state history (the particles live in registers) - come from
  *replays of its prefixes*: ``replay=(seed, u)`` repeats a run's draws exactly (what SMC^2 uses to cut a block at a
  rejuvenation, ``SMC2.fit``), so the first ``j`` moves of the run are a ``j``-move run of the same kernel.  The last
  prefix IS the run (asserted bit for bit);
* ``pf_debug_draw_normals`` dumps the normals the kernel consumed, and the oracle (``oracle/cpu_ref.py`` restating
  ``pyfilter/filters/particle/apf.py:25-46``, ``sisr.py:14-56``, the proposals) is teacher-forced from the kernel's own
  state ``j`` in float64 and in float32: state ``j + 1`` must match particle for particle - bars of ``_compare_step``
  (x 1e-5 of the state's scale, w 2e-5 rel + 2e-4 on identical ancestors, ancestor flips <= 2e-4 N B or 3 x the flips
  between the two oracles, log-likelihood increment 1e-4 + 10 flips / N)."""
import math

import pytest
import torch

from oracle.cases import build_spec, simulate
from pyfilter_amd import ops
from tests.helpers import build_ssm_from_case
from tests.test_production_kernels_gpu import F32, _compare_step, _normals_ref_layout, _oracle_step

pytestmark = pytest.mark.gpu

MODEL_OF = {"lg": "lg1d", "sine": "sine", "ou": "ou_batched", "sv": "sv_batched", "lorenz": "lorenz", "rw2d": "rw2d"}


def _specialised(kind, n):
    """Does a compile-time-specialised instantiation exist for this float32 run (pf_kernels.hip::column_run_impl)?"""
    if kind in ("lg", "sine", "ou", "sv"):
        return True  # scalar kinds: aligned or RAGGED, the 256- or the 1024-thread bound
    if kind == "lorenz":
        return n % 4 == 0  # (the 256- or the 1024-thread bound; N % 4 != 0: the run-time kernel's RAGGED instantiation)
    return False  # D = 2: the run-time kernel


def _cases():
    out = []
    for kind in ("lg", "sine", "ou", "sv", "lorenz"):
        for filt_name in ("sisr", "apf"):
            for prop in ("bootstrap", "lgo"):
                if kind == "sv" and prop == "lgo":
                    continue
                for n in (512, 333, 1502):
                    if kind == "lorenz" and n != 512:
                        continue
                    out.append((kind, filt_name, prop, n))
    # Lorenz under the 1024-thread bound (specialised since round 4); not specialised, float32 all the same: Lorenz of
    # N % 4 != 0 and the D = 2 kernels - for N % 4 != 0 their RAGGED instantiations (four particles per lane for every D)
    out += [("lorenz", "apf", "lgo", 1536), ("lorenz", "sisr", "bootstrap", 1536), ("lorenz", "sisr", "bootstrap", 333),
            ("lorenz", "apf", "lgo", 1502),
            ("rw2d", "apf", "lgo", 512), ("rw2d", "sisr", "bootstrap", 1000), ("rw2d", "sisr", "lgo", 333), ("rw2d", "apf", "bootstrap", 1502)]
    return out


@pytest.mark.parametrize("kind,filt_name,prop,n", _cases())
def test_production_column_kernels_match_oracle_on_their_own_draws(kind, filt_name, prop, n):
    own_draws_check(kind, filt_name, prop, n, 9, 1 if _specialised(kind, n) else 0)


def own_draws_check(kind, filt_name, prop, n, spec_id, fast, bt=None):
    """The procedure of the module docstring for one configuration; ``spec_id`` / ``fast``: what the launch trace must show (9 = the
    column kernel, 10 = the column-cluster kernel of ``tests/test_cluster_route_gpu.py``)."""
    from pyfilter_amd.filters.particle import APF, SISR, proposals

    b, t_len = bt if bt is not None else ((5, 10) if n < 1024 else (3, 7))
    ess = 0.9 if filt_name == "apf" else 0.6
    case = dict(name=f"{kind}_{filt_name}_{prop}_{n}", model=MODEL_OF[kind], filter=filt_name, proposal=prop, N=n, B=b,
                T=t_len, ess_threshold=ess, seed=900 + n, dtypes=("f32",))
    spec64, spec32 = build_spec(case, torch.float64), build_spec(case, F32)
    y = simulate(case, spec64).to(F32)
    y[4] = float("nan")  # a missing observation inside the loop (APF: observed -> unobserved -> observed transitions)
    d, has_event = max(1, spec64.dim), spec64.dim > 0
    gen = torch.Generator().manual_seed(31 + n)
    u = torch.rand((t_len, b), generator=gen, dtype=F32)

    ssm = build_ssm_from_case(case, F32, "cuda")
    cls = APF if filt_name == "apf" else SISR
    filt = cls(ssm, n, proposal={"bootstrap": proposals.Bootstrap, "lgo": proposals.LinearGaussianObservations}[prop](),
               ess_threshold=ess, seed=4321)
    filt.set_batch_shape(torch.Size([b]))
    filt.set_tape(u=u)  # uniforms injected (time-indexed), normals stay Philox: the specialised kernels are selected
    s0 = filt.initialize()

    # ---- the run under test: ONE launch of the column kernel for all t_len moves ----------------------------------------
    full = filt._batch_filter_fused(y.cuda(), s0._restarted())
    torch.cuda.synchronize()
    tr = ops.debug_launch_trace(1)[-1]
    assert tr["SPEC"] == spec_id and tr["tbytes"] == 4 and tr["D"] == d and tr["step"] == 0, tr
    assert tr["FAST"] == fast, (tr, "expected a specialised instantiation" if fast else "expected the run-time kernel")
    seed = filt._last_run["seed_eff"]
    ll_steps = filt._last_run["ll_steps"].cpu()  # (t_len, B): the moves' log-likelihood increments
    z = _normals_ref_layout(filt, t_len, n, b, d, has_event)  # (t_len, N, B, [D]): exactly what the kernel drew

    # ---- its intermediate states: replays of its prefixes (same draws) -------------------------------------------------
    def state_after(j):
        if j == 0:
            return s0
        r = filt._batch_filter_fused(y[:j].cuda(), s0._restarted(), replay=(seed, None))
        trj = ops.debug_launch_trace(1)[-1]
        assert trj["SPEC"] == spec_id and trj["FAST"] == tr["FAST"]
        torch.testing.assert_close(filt._last_run["ll_steps"].cpu(), ll_steps[:j], rtol=0, atol=0)  # the same run, cut
        return r.latest_state

    last = state_after(t_len)
    assert torch.equal(last.timeseries_state.value, full.latest_state.timeseries_state.value), "a replay is not the run"
    assert torch.equal(last.previous_indices, full.latest_state.previous_indices)

    prev = s0
    total_flips = 0
    for j in range(t_len):
        nxt = state_after(j + 1)
        x_in, w_in = prev.timeseries_state.value.cpu(), prev.weights.cpu().clone()
        idx_in = prev.previous_indices.cpu()
        r64 = _oracle_step(spec64, case, y[j], x_in, w_in, idx_in, z[j], u[j], torch.float64)
        r32 = _oracle_step(spec32, case, y[j], x_in, w_in, idx_in, z[j], u[j], F32)
        flips, _ = _compare_step(f"{case['name']} move {j}", nxt.timeseries_state.value.cpu(), nxt.weights.cpu(), ll_steps[j],
                                 nxt.previous_indices.cpu(), r64, r32, n)
        total_flips += flips
        prev = nxt
    # the moment rows of the run: the weighted mean of every state the oracle was just handed (row j + 1 <-> state j + 1)
    W = torch.softmax(torch.nan_to_num(prev.weights.cpu().double(), nan=-math.inf), dim=0)
    xl = prev.timeseries_state.value.cpu().double()
    mean_last = (W.unsqueeze(-1) * xl).sum(0) if has_event else (W * xl).sum(0)
    got = full.filter_means[-1].cpu().double().reshape(mean_last.shape)
    torch.testing.assert_close(got, mean_last, rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(full.loglikelihood.cpu().double().reshape(-1), ll_steps.double().sum(0).reshape(-1), rtol=1e-5, atol=1e-4)
