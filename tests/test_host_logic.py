"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol ``include/pf_amd.h`` declares, the host
logic (parameter packing, layouts, containers, step schedule) behaves, the product refuses to run without a GPU, and
the plain-C oracle agrees with the reference's golden vectors."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def built():
    import __graft_entry__ as ge

    ge.build()


def test_library_exports_every_declared_symbol():
    from pyfilter_amd import _lib

    header = open(os.path.join(ROOT, "include", "pf_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(pf_[a-z_0-9]+)\s*\(", header))
    assert declared, "no declarations parsed"
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in pf_amd.h but not exported by libpfamd.so"
    assert declared <= set(_lib.EXPORTS), declared - set(_lib.EXPORTS)
    assert b"gfx950" in lib.pf_version()
    assert lib.pf_error_string(-2) == b"workspace too small"
    n = C.c_size_t(0)
    assert lib.pf_workspace_bytes(1 << 20, 1, 1, C.byref(n)) == 0 and n.value > 0
    assert lib.pf_workspace_bytes(0, 1, 1, C.byref(n)) == _lib_einval()


def _lib_einval():
    return -1


def test_argument_validation_without_gpu():
    """Bad arguments are rejected before anything is launched (safe to call on a CPU-only box)."""
    from pyfilter_amd import _lib

    lib = _lib.load()
    assert lib.pf_normalize(None, None, None, None, 10, 1, 0, None, 0, None) == -1
    assert lib.pf_filter_run(None, 0, 1, 0, None) == -1
    m = _lib.PfModel()
    m.hid_kind, m.obs_kind, m.dim, m.obs_dim, m.params = 3, 0, 1, 1, 1  # Lorenz needs D = 3
    assert lib.pf_pre_weight(C.byref(m), 0, 1, 1, 1, 1, 8, 1, 0, None) == -3


def test_no_cpu_fallback():
    import pyfilter_amd
    from pyfilter_amd import _lib

    with pytest.raises(_lib.PfAmdError):
        pyfilter_amd.utils.normalize(torch.zeros(8))
    with pytest.raises(_lib.PfAmdError):
        pyfilter_amd.resampling.systematic(torch.zeros(8))


def test_pack_params_layout():
    from pyfilter_amd import timeseries as ts
    from pyfilter_amd.timeseries import models

    t = torch.tensor
    kappa, gamma, sigma = t([0.1, 0.2, 0.3]), t([1.0, 1.1, 1.2]), t([0.5, 0.6, 0.7])
    ssm = ts.LinearStateSpaceModel(models.OrnsteinUhlenbeck(kappa, gamma, sigma, dt=1.0), (t(2.0), t(0.25)))
    rows = models.pack_params(ssm, 3, torch.float64, "cpu")
    assert rows.shape == (3, 4 * 1 + 1 + 2)  # [hp0 hp1 hp2 hp3 | A | b | s]
    assert torch.equal(rows[:, 0], kappa.double()) and torch.equal(rows[:, 2], sigma.double())
    assert torch.equal(rows[:, 4], torch.full((3,), 2.0, dtype=torch.float64))
    assert torch.equal(rows[:, 5], torch.zeros(3, dtype=torch.float64)) and torch.equal(rows[:, 6], torch.full((3,), 0.25, dtype=torch.float64))
    k = ssm.kernel_kind
    assert (k.hid_kind, k.obs_kind, k.dim, k.obs_dim) == (4, 0, 1, 1)

    hidden = models.Lorenz63(t(10.0), t(28.0), t(8.0 / 3.0), t(1.0))
    a = t([[0.8, 0.0, 0.0], [0.0, 0.0, 0.8]])
    ssm3 = ts.LinearStateSpaceModel(hidden, (a, t([0.0]), t([0.3])), torch.Size([2]))
    rows = models.pack_params(ssm3, 2, torch.float32, "cpu")
    assert rows.shape == (2, 4 * 3 + 2 * 3 + 2 * 2)
    assert torch.equal(rows[0, 12:18], a.reshape(-1)) and torch.allclose(rows[1, 20:22], t([0.3, 0.3]))
    assert ssm3.kernel_kind.dim == 3 and ssm3.kernel_kind.obs_dim == 2 and abs(ssm3.kernel_kind.inc_scale - 0.1) < 1e-12

    sv = models.StochasticVolatilityModel(models.Verhulst(t(0.05), t(1.0), t(0.1), dt=0.2), t(0.0))
    assert sv.kernel_kind.obs_kind == 1
    with pytest.raises(Exception):
        models.pack_params(ts.StateSpaceModel(ts.AffineProcess(lambda x, s: (x.value, s), (t(1.0),), None, None), None, ()), 1, torch.float32, "cpu")


def test_layout_views_round_trip_without_copies():
    from pyfilter_amd import ops

    soa = torch.arange(2 * 3 * 5, dtype=torch.float32).reshape(2, 3, 5)  # (D, B, N)
    v = ops.from_soa(soa, True, True)
    assert v.shape == (5, 3, 2) and ops.to_soa(v, True, True).data_ptr() == soa.data_ptr()
    cols = torch.arange(12, dtype=torch.float32).reshape(3, 4)
    w = ops.from_cols(cols, True)
    assert w.shape == (4, 3) and ops.to_cols(w).data_ptr() == cols.data_ptr()
    one = torch.zeros(1, 4)
    assert ops.from_cols(one, False).shape == (4,)
    s1 = torch.zeros(1, 1, 6)
    assert ops.from_soa(s1, False, False).shape == (6,) and ops.to_soa(ops.from_soa(s1, False, False), False, False).data_ptr() == s1.data_ptr()


def test_tensor_container_state_dict_keys():
    from pyfilter_amd.container import TensorContainer, make_dequeue

    tc = TensorContainer()
    tc.make_deque("filter_means", maxlen=True)
    tc.make_deque("last", maxlen=False)
    tc.make_deque("five", maxlen=5)
    for i in range(7):
        for k in ("filter_means", "last", "five"):
            tc[k].append(torch.full((2,), float(i)))
    sd = tc.state_dict()
    assert set(sd) == {"tensor_deque_None__filter_means", "tensor_deque_1__last", "tensor_deque_5__five"}
    assert sd["tensor_deque_None__filter_means"].shape == (7, 2) and sd["tensor_deque_1__last"].shape == (1, 2)
    assert sd["tensor_deque_5__five"][0, 0] == 2.0
    tc2 = TensorContainer()
    tc2.load_state_dict(dict(sd))
    assert torch.equal(tc2.get_as_tensor("five"), tc.get_as_tensor("five")) and tc2["five"].maxlen == 5
    assert make_dequeue(False).maxlen == 1 and make_dequeue(True).maxlen is None


def test_api_surface_matches_reference_names():
    from pyfilter_amd.filters import BaseFilter, FilterResult
    from pyfilter_amd.filters.particle import APF, SISR, ParticleFilter, proposals
    from pyfilter_amd.resampling import multinomial, systematic  # noqa: F401
    from pyfilter_amd.utils import get_ess, normalize  # noqa: F401

    for m in ("set_batch_shape", "initialize", "predict", "correct", "filter", "batch_filter", "copy", "increase_particles"):
        assert hasattr(SISR, m) and hasattr(APF, m), m
    for m in ("filter_means", "filter_variance", "loglikelihood", "latest_state", "states", "state_dict",
              "load_state_dict", "resample", "exchange", "copy"):
        assert hasattr(FilterResult, m), m
    for cls in (proposals.Bootstrap, proposals.LinearGaussianObservations):
        for m in ("set_model", "sample_and_weight", "pre_weight", "copy"):
            assert hasattr(cls, m)
    assert issubclass(SISR, ParticleFilter) and issubclass(ParticleFilter, BaseFilter)
    f = SISR(lambda ctx: None, 100, ess_threshold=0.5)
    f.set_batch_shape(torch.Size([7]))
    assert f.particles == torch.Size([100, 7]) and f._resample_threshold == 50.0
    f.increase_particles(2)
    assert f.particles == torch.Size([200, 7]) and f._resample_threshold == 100.0
    g = f.copy()
    assert g.particles == f.particles and g._resample_threshold == 100.0 * 200  # the reference's copy() quirk
    with pytest.raises(NotImplementedError):
        f.set_batch_shape(torch.Size([2, 3]))


# ---- the plain-C oracle against the reference's golden vectors -----------------------------------------------------
@pytest.mark.parametrize("dt,ct,suf", [("f32", np.float32, "f32"), ("f64", np.float64, "f64")])
def test_c_oracle_against_reference_golden(dt, ct, suf):
    so = C.CDLL(os.path.join(ROOT, "oracle", "_build", "libpforacle.so"))
    with np.load(os.path.join(ROOT, "tests", "golden", f"primitives_{dt}.npz")) as f:
        g = {k: f[k] for k in f.files}
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    for nm in "abc":
        lw = np.ascontiguousarray(g[f"norm_{nm}_in"].T)  # (B, N) column layout
        b, n = lw.shape
        W = np.empty_like(lw)
        getattr(so, f"oracle_normalize_{suf}")(ptr(lw), ptr(W), C.c_int64(n), C.c_int64(b))
        assert np.array_equal(lw.T, g[f"norm_{nm}_inplace"], equal_nan=True)  # in-place sanitisation: exact
        np.testing.assert_allclose(W.T, g[f"norm_{nm}_W"], rtol=1e-12 if dt == "f64" else 5e-5, atol=0)
        # sequential systematic walk on the reference's own W and u: exact ancestors
        Wref = np.ascontiguousarray(g[f"norm_{nm}_W"].T)
        ok = ~np.isnan(Wref).any(axis=1)
        u = np.ascontiguousarray(g[f"norm_{nm}_u"].reshape(-1))
        idx = np.empty((b, n), dtype=np.int64)
        getattr(so, f"oracle_systematic_{suf}")(ptr(Wref), ptr(u), ptr(idx), C.c_int64(n), C.c_int64(b))
        assert np.array_equal(idx[ok].T, g[f"norm_{nm}_idx"][:, ok])


def test_ctypes_structs_match_the_c_header(tmp_path):
    """The ctypes mirrors of ``pf_model`` / ``pf_filter_args`` have the size and field offsets the C compiler gives the
    structs of ``include/pf_amd.h`` (an ABI drift between header and binding would corrupt every pointer after it)."""
    import ctypes as C
    import subprocess

    from pyfilter_amd import _lib as L

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fields = {"pf_model": [f[0] for f in L.PfModel._fields_], "pf_filter_args": [f[0] for f in L.PfFilterArgs._fields_]}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{root}/include/pf_amd.h"', "int main(void) {"]
    for st, names in fields.items():
        lines.append(f'  printf("{st} %zu\\n", sizeof({st}));')
        for n in names:
            lines.append(f'  printf("{st}.{n} %zu\\n", offsetof({st}, {n}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "abi.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-std=c11", str(src), "-o", str(exe)])
    out = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for st, cls in (("pf_model", L.PfModel), ("pf_filter_args", L.PfFilterArgs)):
        assert int(out[st]) == C.sizeof(cls), (st, out[st], C.sizeof(cls))
        for n in fields[st]:
            assert int(out[f"{st}.{n}"]) == getattr(cls, n).offset, (st, n, out[f"{st}.{n}"], getattr(cls, n).offset)


# ---- result containers / drivers (pure host logic, CPU tensors) ----------------------------------------------------
class _ToyState(dict):
    """A Correction-shaped object for the host-logic tests: mean / variance / ll tensors over a batch of B filters."""

    def __init__(self, t, mean, var, ll):
        super().__init__()
        self.t, self.mean, self.var, self.ll = t, mean, var, ll

    def get_mean(self):
        return self.mean

    def get_variance(self):
        return self.var

    def get_loglikelihood(self):
        return self.ll

    def resample(self, indices):
        self.mean, self.var = self.mean[indices], self.var[indices]

    def exchange(self, other, mask):
        self.mean[mask], self.var[mask] = other.mean[mask], other.var[mask]

    def state_dict(self):
        return {"mean": self.mean}

    def load_state_dict(self, sd):
        self.mean = sd["mean"]


def _toy_states(n, b=3, d=2, seed=0):
    g = torch.Generator().manual_seed(seed)
    return [_ToyState(t, torch.randn(b, d, generator=g), torch.rand(b, d, generator=g), torch.full((b,), 0.5 * t)) for t in range(n)]


@pytest.mark.parametrize("record_moments,expect", [(True, 9), (False, 1), (4, 4), (1, 1)])
def test_filter_result_moment_log_window(record_moments, expect):
    """The moment series behaves like the reference's deque(maxlen) of per-step tensors: rows, order, bound."""
    from pyfilter_amd.filters import FilterResult

    states = _toy_states(9)
    res = FilterResult(states[0], False, record_moments)
    for s in states[1:]:
        res.append(s)
    want_m = torch.stack([s.mean for s in states[-expect:]])
    want_v = torch.stack([s.var for s in states[-expect:]])
    assert res.filter_means.shape == (expect, 3, 2)
    assert torch.equal(res.filter_means, want_m) and torch.equal(res.filter_variance, want_v)
    # bulk rows (the fused kernels' hand-over) join the same window
    more_m, more_v = torch.randn(6, 3, 2), torch.rand(6, 3, 2)
    res._extend_fused(more_m, more_v, torch.zeros(3), states[-1])
    full_m = torch.cat([torch.stack([s.mean for s in states]), more_m])
    keep = full_m.shape[0] if record_moments is True else expect
    assert torch.equal(res.filter_means, full_m[-keep:])
    assert res.latest_state is states[-1] and len(res.states) == 1
    # log-likelihood: running total on the (aliased) tensor of the initial state (result.py:34)
    assert res.loglikelihood is states[0].ll


def test_filter_result_resample_exchange_and_wire_format():
    from pyfilter_amd.filters import FilterResult

    sa, sb = _toy_states(6, seed=1), _toy_states(6, seed=2)
    ra, rb = FilterResult(sa[0], True, True), FilterResult(sb[0], True, True)
    for x, y in zip(sa[1:], sb[1:]):
        ra.append(x)
        rb.append(y)
    ma, mb = ra.filter_means.clone(), rb.filter_means.clone()
    mask = torch.tensor([True, False, True])
    ra.exchange(rb, mask)
    want = ma.clone()
    want[:, mask] = mb[:, mask]
    assert torch.equal(ra.filter_means, want) and torch.equal(rb.filter_means, mb)
    idx = torch.tensor([2, 2, 0])
    ll_before = ra.loglikelihood.clone()
    ra.resample(idx)
    assert torch.equal(ra.filter_means, want[:, idx]) and torch.equal(ra.loglikelihood, ll_before[idx])
    ra.resample(torch.tensor([1, 0, 0]), entire_history=False)  # states + ll only
    assert torch.equal(ra.filter_means, want[:, idx])
    # wire format: the reference's keys (container.py:113-139, result.py:135-154)
    sd = ra.state_dict()
    assert list(sd) == ["tensor_tuples", "state", "log_likelihood"]
    assert list(sd["tensor_tuples"]) == ["tensor_deque_None__filter_means", "tensor_deque_None__filter_variances"]
    assert sd["tensor_tuples"]["tensor_deque_None__filter_means"].shape == (6, 3, 2)
    fresh = FilterResult(_toy_states(1)[0], False, True)
    fresh.load_state_dict(sd)
    assert torch.equal(fresh.filter_means, ra.filter_means) and torch.equal(fresh.filter_variance, ra.filter_variance)
    assert fresh.loglikelihood is sd["log_likelihood"]
    tt = ra.tensor_tuples  # the reference's container view
    assert set(tt.keys()) == {"filter_means", "filter_variances"} and len(tt["filter_means"]) == 6
    bounded = FilterResult(_toy_states(1)[0], False, 5)
    assert "tensor_deque_5__filter_means" in bounded.state_dict()["tensor_tuples"]


def test_unbatched_scalar_moment_rows_keep_their_shape():
    from pyfilter_amd.filters import FilterResult

    st = [_ToyState(t, torch.tensor(float(t)), torch.tensor(0.1 * t), torch.tensor(0.0)) for t in range(4)]
    res = FilterResult(st[0], False, True)
    for s in st[1:]:
        res.append(s)
    assert res.filter_means.shape == (4,) and res.filter_means.tolist() == [0.0, 1.0, 2.0, 3.0]
    st = [_ToyState(t, torch.full((1,), float(t)), torch.zeros(1), torch.tensor(0.0)) for t in range(3)]
    res = FilterResult(st[0], False, True)
    res._extend_fused(torch.tensor([[1.0], [2.0]]), torch.zeros(2, 1), torch.tensor(0.0), st[-1])
    assert res.filter_means.shape == (3, 1)


def test_move_schedule():
    from pyfilter_amd.filters.schedule import expand, unobserved_moves_before

    assert [unobserved_moves_before(t, 3) for t in range(7)] == [0, 2, 1, 0, 2, 1, 0]
    assert expand(0, 3, 1) == ([0, 1, 2], [0, 1, 2])
    s = expand(1, 2, 3)
    assert s.source == [-1, -1, 0, -1, -1, 1] and s.rows == [2, 5] and s.moves == 6
    assert expand(0, 0, 2).moves == 0


def test_generic_driver_walks_the_schedule():
    """``BaseFilter.filter`` / ``batch_filter``: unobserved sub-steps before every observation, a propagate-only move for
    an all-NaN observation, intermediary states recorded only on request (filters/base.py:188-221)."""
    from pyfilter_amd import timeseries as ts
    from pyfilter_amd.filters.base import BaseFilter
    from pyfilter_amd.timeseries import models

    class Pred:
        def __init__(self, state):
            self.state = state

        def get_timeseries_state(self):
            return self.state.tsx

        def create_state_from_prediction(self, model):
            return Toy(self.state.tsx.propagate_from(values=self.state.tsx.value), "propagated")

    class Toy(_ToyState):
        def __init__(self, tsx, kind):
            super().__init__(int(tsx.time_index), tsx.value.clone(), torch.zeros(()), torch.zeros(()))
            self.tsx, self.kind = tsx, kind

        def get_timeseries_state(self):
            return self.tsx

    class F(BaseFilter):
        def initialize(self):
            return Toy(ts.TimeseriesState(0, torch.tensor(0.0), torch.Size([])), "init")

        def predict(self, state):
            return Pred(state)

        def correct(self, y, prediction):
            x = prediction.state.tsx
            return Toy(x.propagate_from(values=torch.as_tensor(y, dtype=torch.float32)), "corrected")

    hidden = models.AR(torch.tensor(0.0), torch.tensor(0.9), torch.tensor(0.1))
    ssm = ts.LinearStateSpaceModel(hidden, (torch.tensor(1.0), torch.tensor(0.1)), observe_every_step=3)
    y = torch.tensor([1.0, float("nan"), 3.0])
    for inter, kinds in ((False, ["init", "corrected", "propagated", "corrected"]),
                         (True, ["init", "corrected", "propagated", "propagated", "propagated", "propagated", "propagated", "corrected"])):
        f = F(ssm, record_states=True, record_intermediary_states=inter)
        res = f.batch_filter(y, bar=False)
        assert [s.kind for s in res.states] == kinds
        assert int(res.latest_state.tsx.time_index) == 7  # 0 -> 1, then 2 sub-steps + 1 move per further observation
        assert res.filter_means.shape[0] == len(kinds)
    with pytest.raises(NotImplementedError):
        F(ssm, nan_strategy="drop")
    with pytest.raises(ValueError):
        F(42)


def test_standalone_cpp_program_compiles_and_links_against_the_header(tmp_path):
    """``tests/c_abi/standalone.cpp`` (the C ABI from plain C++: no Python, no PyTorch) builds against ``include/pf_amd.h`` and
    links ``libpfamd.so`` + the HIP runtime here; it runs in the ``-m gpu`` suite."""
    import shutil
    import subprocess

    import __graft_entry__ as ge

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    if shutil.which("g++") is None or not os.path.exists(os.path.join(rocm, "lib", "libamdhip64.so")):
        pytest.skip("g++ / the HIP runtime are not installed here")
    ge.build()
    cmd = ["g++", "-O1", "-D__HIP_PLATFORM_AMD__", f"-I{rocm}/include", f"-I{root}/include",
           os.path.join(root, "tests", "c_abi", "standalone.cpp"), "-o", str(tmp_path / "standalone"),
           f"-L{root}/pyfilter_amd", "-lpfamd", f"-L{rocm}/lib", "-lamdhip64"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]


def test_user_defined_affine_processes_get_the_fused_kernel_kind():
    """A lambda-defined ``AffineProcess`` (the reference's plug-in seam) under a ``LinearStateSpaceModel`` is given the
    ``PF_HID_USER_AFFINE`` kernel kind when its increments are centred Gaussians of one scale - and only then."""
    from torch.distributions import Exponential, Independent, Normal

    from pyfilter_amd import _lib as L, timeseries as ts
    from pyfilter_amd.timeseries.models import pack_params

    f = lambda x, a, s: (a * x.value, s)  # noqa: E731
    init = lambda a, s: Normal(torch.tensor(0.0), torch.tensor(1.0))  # noqa: E731
    hid = ts.AffineProcess(f, (0.9, 0.3), Normal(torch.tensor(0.0), torch.tensor(0.5)), init)
    ssm = ts.LinearStateSpaceModel(hid, (1.0, 0.1))
    k = ssm.kernel_kind
    assert k is not None and k.is_user and k.hid_kind == L.HID_USER_AFFINE and k.dim == 1 and k.inc_scale == 0.5 and k.obs_kind == L.OBS_LINEAR
    rows = pack_params(ssm, 3, torch.float64, torch.device("cpu"))
    assert rows.shape == (3, 4 + 1 + 2) and (rows[:, :4] == 0).all() and torch.allclose(rows[0, 4:], torch.tensor([1.0, 0.0, 0.1], dtype=torch.float64))
    # a vector state with independent increments of one scale
    init3 = lambda *_: Independent(Normal(torch.zeros(3), torch.ones(3)), 1)  # noqa: E731
    hid3 = ts.AffineProcess(lambda x, s: (x.value, s), (0.2,), Independent(Normal(torch.zeros(3), torch.full((3,), 0.1)), 1), init3)
    k3 = ts.LinearStateSpaceModel(hid3, (torch.eye(3)[:2], torch.zeros(2), torch.full((2,), 0.3)), torch.Size([2])).kernel_kind
    assert k3 is not None and k3.is_user and k3.dim == 3 and k3.obs_dim == 2 and abs(k3.inc_scale - 0.1) < 1e-7
    # not Gaussian / not centred / component-wise scales: the step-by-step route keeps those
    for inc in (Exponential(torch.tensor(1.0)), Normal(torch.tensor(0.2), torch.tensor(1.0))):
        assert ts.LinearStateSpaceModel(ts.AffineProcess(f, (0.9, 0.3), inc, init), (1.0, 0.1)).kernel_kind is None
    uneven = Independent(Normal(torch.zeros(3), torch.tensor([0.1, 0.2, 0.1])), 1)
    assert ts.LinearStateSpaceModel(ts.AffineProcess(lambda x, s: (x.value, s), (0.2,), uneven, init3),
                                    (torch.eye(3)[:2], torch.zeros(2), torch.full((2,), 0.3)), torch.Size([2])).kernel_kind is None
    # built-in kinds are not "user"
    from pyfilter_amd.timeseries import models

    assert not ts.LinearStateSpaceModel(models.AR(0.0, 0.9, 0.1), (1.0, 0.1)).kernel_kind.is_user


def test_optimal_proposal_innovation_form_equals_the_precision_form():
    """``proposals/linear.py::_ObservationUpdate`` (the step-by-step route's optimal proposal for user-defined affine models,
    written in innovation form) against a per-particle brute force of the textbook precision form - posterior mean /
    covariance and the first-stage marginal - for scalar / vector states and observations, shared and per-filter ``A``."""
    from torch.distributions import MultivariateNormal

    from pyfilter_amd.filters.particle.proposals.linear import _ObservationUpdate

    class Obj:
        pass

    def check(vx, vy, batched_a, N=5, B=3, D=3, O=2):
        f64 = torch.float64
        g = torch.Generator().manual_seed(1)
        ev = (D,) if vx else ()
        m, xcur = torch.randn((N, B) + ev, generator=g, dtype=f64), torch.randn((N, B) + ev, generator=g, dtype=f64)
        h = 0.3 + torch.rand((N, B) + ev, generator=g, dtype=f64)
        shape_a = ((O, D) if vy else (D,)) if vx else ((O,) if vy else ())
        a = torch.randn(((B,) if batched_a else ()) + shape_a, generator=g, dtype=f64)
        oe = (O,) if vy else ()
        b, y = torch.randn(oe, generator=g, dtype=f64), torch.randn(oe, generator=g, dtype=f64)
        s = 0.2 + torch.rand(oe, generator=g, dtype=f64)
        model = Obj()
        model.hidden = Obj()
        model.hidden.n_dim, model.n_dim, model.parameters = int(vx), int(vy), (a, b, s)
        up = _ObservationUpdate(model, h)
        post, lm = up.posterior(y, m), up.log_marginal(y, xcur)
        assert post.batch_shape == (N, B) and lm.shape == (N, B)
        dd, oo = (D if vx else 1), (O if vy else 1)
        for n in range(N):
            for k in range(B):
                A = (a[k] if batched_a else a).reshape(oo, dd)
                Hm, R = torch.diag(h[n, k].reshape(-1) ** 2), torch.diag(s.reshape(-1).expand(oo) ** 2)
                cov = torch.linalg.inv(torch.linalg.inv(Hm) + A.T @ torch.linalg.inv(R) @ A)
                mean = cov @ (torch.linalg.inv(Hm) @ m[n, k].reshape(-1) + A.T @ torch.linalg.inv(R) @ (y - b).reshape(-1))
                torch.testing.assert_close(post.mean[n, k].reshape(-1), mean, rtol=1e-10, atol=1e-12)
                got_cov = post.covariance_matrix[n, k] if vx else (post.scale[n, k] ** 2).reshape(1, 1)
                torch.testing.assert_close(got_cov, cov, rtol=1e-9, atol=1e-12)
                ref = MultivariateNormal(b.reshape(-1).expand(oo) + A @ xcur[n, k].reshape(-1),
                                         covariance_matrix=R + A @ Hm @ A.T).log_prob(y.reshape(-1).expand(oo))
                torch.testing.assert_close(lm[n, k], ref, rtol=1e-10, atol=1e-12)

    for vx in (False, True):
        for vy in (False, True):
            for batched_a in (False, True):
                check(vx, vy, batched_a)


def test_theta_particles_carry_their_derived_quantities_through_every_move():
    """``ThetaParticles`` keeps the stacked unconstrained values and the summed unconstrained log priors of its current
    values and carries them through ``unstack_parameters`` / ``exchange`` / ``resample`` / ``like`` (SMC^2's rejuvenation
    asks for them several times per move): at every point they equal a fresh evaluation through torch.distributions."""
    from torch.distributions import Beta, Exponential, LogNormal, Normal, Uniform

    from pyfilter_amd.inference.parameters import ThetaParticles

    f64 = torch.float64
    th = ThetaParticles({"a": Exponential(10.0), "b": Normal(0.0, 1.0), "c": LogNormal(-2.0, 1.0), "d": Beta(2.0, 3.0),
                         "e": Uniform(0.2, 0.9)}, 17, "cpu", f64)
    th.initialize_parameters(torch.Generator().manual_seed(1))
    fresh = lambda t: sum(p.eval_prior(t[n], False) for n, p in t.priors.items())  # noqa: E731
    fresh_u = lambda t: torch.cat([p.get_unconstrained(t[n]).reshape(17, -1) for n, p in t.priors.items()], dim=-1)  # noqa: E731
    tol = dict(rtol=1e-10, atol=1e-10)
    torch.testing.assert_close(th.eval_priors(False), fresh(th), **tol)          # (no stack at hand: the plain evaluation)
    u = th.stack_parameters(False)
    th._cache.pop("prior_u")
    torch.testing.assert_close(th.eval_priors(False), fresh(th), **tol)          # (from the stacked values)
    other = th.like()
    g = torch.Generator().manual_seed(2)
    rv = u + 0.1 * torch.randn(u.shape, generator=g, dtype=f64)
    other.unstack_parameters(rv, constrained=False)
    torch.testing.assert_close(other.eval_priors(False), fresh(other), **tol)
    torch.testing.assert_close(other.stack_parameters(False), fresh_u(other), **tol)
    th.exchange(other, torch.rand(17, generator=g) < 0.5)
    torch.testing.assert_close(th.eval_priors(False), fresh(th), **tol)
    torch.testing.assert_close(th.stack_parameters(False), fresh_u(th), **tol)
    th.resample(torch.randint(0, 17, (17,), generator=g))
    torch.testing.assert_close(th.eval_priors(False), fresh(th), **tol)
    torch.testing.assert_close(th.stack_parameters(False), fresh_u(th), **tol)
    th.initialize_parameters(torch.Generator().manual_seed(3))
    assert not th._cache


def test_hints_cluster_rule_mirrors_the_library():
    """``HINTS.cluster_takes`` is the Python statement of ``pf_kernels.hip::cluster_eligible`` (which runs take the column-cluster
    kernel, i.e. the lean single-launch driver): 2 049 .. 16 384 particles, N % 4 == 0, systematic, two launches' worth of member
    workgroups under the default hints, any batch the workspace reserves records for under the tests' routes."""
    from pyfilter_amd.hints import ROUTE_CLUSTER, ROUTE_CLUSTER_ALWAYS, ROUTE_PER_STEP, RunHints

    h = RunHints()
    assert h.kernel_route() == ROUTE_CLUSTER
    assert h.cluster_takes(8192, 128) and h.cluster_takes(8192, 256) and not h.cluster_takes(8192, 257)
    assert h.cluster_takes(2052, 682) and not h.cluster_takes(2052, 683)          # 3 members per filter
    assert not h.cluster_takes(2048, 1) and not h.cluster_takes(16388, 1) and not h.cluster_takes(8190, 1)
    assert not h.cluster_takes(8192, 4, resampler_systematic=False)
    h.column_max_n = 4096                                                            # the column kernel takes precedence up to its bound
    assert not h.cluster_takes(4096, 4) and h.cluster_takes(4100, 4)
    h.column_max_n = 0
    h.cluster = False
    assert h.kernel_route() == 0 and not h.cluster_takes(8192, 4)
    h.cluster, h.route = True, ROUTE_PER_STEP
    assert not h.cluster_takes(8192, 4)
    h.route = ROUTE_CLUSTER_ALWAYS
    assert h.cluster_takes(8192, 1024) and not h.cluster_takes(8192, 1025)         # 8 192 member workgroups: the record space reserved
    m = RunHints().apply_mapping({"PF_NO_CLUSTER": "1"})
    assert m.kernel_route() == 0
    assert RunHints().apply_mapping({"PF_CLUSTER": "1"}).kernel_route() == ROUTE_CLUSTER_ALWAYS


@pytest.mark.parametrize("maxlen", [None, 1, 2, 5, 70])
@pytest.mark.parametrize("batched", [False, True])
def test_moment_log_remembers_single_rows_and_writes_them_when_somebody_looks(maxlen, batched):
    """``MomentLog.append`` only remembers a state's (mean, variance) tensors (no launch per online move); any reader - the
    series, ``rows``, ``extend``, a whole-filter gather - must see exactly what a deque of that ``maxlen`` would hold
    (``pyfilter/container.py:10-18`` + ``result.py:119-133``)."""
    from collections import deque

    from pyfilter_amd.filters.result import MomentLog

    g = torch.Generator().manual_seed(3)
    shape = (4, 2) if batched else (2,)
    log, ref = MomentLog(maxlen), deque(maxlen=maxlen)
    import random

    rnd = random.Random(11)
    for i in range(400):
        op = rnd.random()
        if op < 0.8:
            m, v = torch.randn(shape, generator=g), torch.rand(shape, generator=g)
            log.append(m, v, batched)
            ref.append((m, v))
        elif op < 0.9 and log._buf_ is not None:
            k = rnd.randint(1, 7)
            ms, vs = torch.randn((k,) + shape, generator=g), torch.rand((k,) + shape, generator=g)
            log.extend(ms, vs)
            ref.extend(zip(ms.unbind(0), vs.unbind(0)))
        elif op < 0.95 and batched and len(ref):
            idx = torch.randint(0, shape[0], (shape[0],), generator=g)
            log.gather_filters(idx)
            ref = deque(((m[idx], v[idx]) for m, v in ref), maxlen=maxlen)
        else:
            assert log.rows == len(ref)
            if len(ref):
                assert torch.equal(log.means(), torch.stack([m for m, _ in ref]))
                assert torch.equal(log.variances(), torch.stack([v for _, v in ref]))
    assert log.rows == len(ref)
    assert torch.equal(log.means(), torch.stack([m for m, _ in ref]))
    assert torch.equal(log.variances(), torch.stack([v for _, v in ref]))

