"""
TEST INFRASTRUCTURE — the golden-fixture case table shared by ``oracle/make_golden.py`` (generator, this container
only) and ``tests/`` (consumers; they only need the spec, the inputs come from the ``.npz``).

Model constants follow SURVEY.md §8(c)/(d): AR(1) of tests/filters/models.py:10-15, the README sine diffusion,
the Verhulst SV model of examples/stochastic-volatility.ipynb, Lorenz-63 of examples/lorenz.ipynb and the OU model
of tests/inference/models.py:12-19, and the reference's own 2-D acceptance model - the random walk with sigma = (0.05, 0.1),
A = I2, s = 0.15 of tests/filters/models.py:28-52 (``rw2d``; B in {1, 3} and 10 % NaN rows as tests/filters/test_particle.py:44-63
runs it).

Round 5 - the observation shapes and schedules the kernels accept beyond those: a SCALAR observation of a vector state
(``event_shape = Size([])`` with ``a`` of shape ``(D,)``: the ``obs_is_1d`` branch of proposals/utils.py:243-260; ``lorenz_s``,
``rw2d_s``) and the same observation declared as a vector of length one (``event_shape = Size([1])``, ``a`` of shape ``(1, D)``:
``lorenz_o1``, ``rw2d_o1`` - the only way the reference's APF + LinearGaussianObservations runs on such a model: its
``pre_weight`` mixes ``(N, B, D)`` and ``(N, B)`` tensors for ``obs_is_1d`` and D > 1, proposals/linear.py:79-81), D = 3 / O = 3
with a dense ``A``, a non-zero offset and three distinct noise scales (``lorenz_o3``), per-filter ``sigma / A / b / s`` rows on a
vector model (``rw2d_theta``), and ``observe_every_step > 1`` (filters/base.py:196-221; ``*_oes3``).
"""
import math

import torch

from . import models as M

_LORENZ_INIT = ([-5.91652, -5.52332, 24.5723], [math.sqrt(10.0)] * 3)
_LORENZ_A = [[0.8, 0.0, 0.0], [0.0, 0.0, 0.8]]

CASES = [
    # name, model, filter, proposal, N, B, T
    dict(name="lg1d_sisr_boot", model="lg1d", filter="sisr", proposal="bootstrap", N=1000, B=1, T=25,
         ess_threshold=0.9, seed=101, dtypes=("f64", "f32")),
    dict(name="lg1d_apf_lgo", model="lg1d", filter="apf", proposal="lgo", N=256, B=3, T=20,
         ess_threshold=0.9, seed=102, dtypes=("f64", "f32")),
    dict(name="sine_apf_lgo", model="sine", filter="apf", proposal="lgo", N=512, B=2, T=20,
         ess_threshold=0.9, seed=103, dtypes=("f64", "f32")),
    dict(name="sine_sisr_lgo", model="sine", filter="sisr", proposal="lgo", N=256, B=3, T=20,
         ess_threshold=0.5, seed=104, dtypes=("f64",)),
    dict(name="sine_apf_boot_nan", model="sine", filter="apf", proposal="bootstrap", N=256, B=2, T=20,
         ess_threshold=0.9, seed=105, nan_steps=(3, 4, 11), dtypes=("f64",)),
    dict(name="sine_sisr_boot_nan", model="sine", filter="sisr", proposal="bootstrap", N=256, B=3, T=20,
         ess_threshold=0.7, seed=106, nan_steps=(2, 9), dtypes=("f64",)),
    dict(name="sv_apf_boot", model="sv_batched", filter="apf", proposal="bootstrap", N=256, B=4, T=20,
         ess_threshold=0.9, seed=107, dtypes=("f64", "f32")),
    dict(name="sv_sisr_boot", model="sv_batched", filter="sisr", proposal="bootstrap", N=256, B=4, T=20,
         ess_threshold=0.6, seed=108, dtypes=("f64",)),
    dict(name="lorenz_sisr_boot", model="lorenz", filter="sisr", proposal="bootstrap", N=256, B=2, T=15,
         ess_threshold=0.9, seed=109, dtypes=("f64", "f32")),
    dict(name="lorenz_apf_lgo", model="lorenz", filter="apf", proposal="lgo", N=128, B=2, T=15,
         ess_threshold=0.9, seed=110, dtypes=("f64",)),
    dict(name="ou_sisr_lgo_theta", model="ou_batched", filter="sisr", proposal="lgo", N=128, B=3, T=20,
         ess_threshold=0.8, seed=111, dtypes=("f64",)),
    dict(name="ou_apf_boot_theta", model="ou_batched", filter="apf", proposal="bootstrap", N=128, B=3, T=20,
         ess_threshold=0.9, seed=112, dtypes=("f64",)),
    # D = 2 / O = 2: the reference's 2-D random walk (tests/filters/models.py:28-52), its MVN / Cholesky LGO path included
    dict(name="rw2d_sisr_boot", model="rw2d", filter="sisr", proposal="bootstrap", N=512, B=3, T=25,
         ess_threshold=0.9, seed=113, dtypes=("f64", "f32")),
    dict(name="rw2d_apf_lgo", model="rw2d", filter="apf", proposal="lgo", N=256, B=3, T=25,
         ess_threshold=0.9, seed=114, nan_steps=(4, 13, 14), dtypes=("f64", "f32")),
    dict(name="rw2d_sisr_lgo", model="rw2d", filter="sisr", proposal="lgo", N=300, B=1, T=20,
         ess_threshold=0.5, seed=115, dtypes=("f64",)),
    dict(name="rw2d_apf_boot", model="rw2d", filter="apf", proposal="bootstrap", N=333, B=2, T=20,
         ess_threshold=0.9, seed=116, nan_steps=(7,), dtypes=("f64",)),
    # ---- round 5 ---------------------------------------------------------------------------------------------------
    # D = 3 / scalar observation (event_shape = Size([])): y = 0.8 x1 + 0.3 x3 + s v
    dict(name="lorenz_s_sisr_lgo", model="lorenz_s", filter="sisr", proposal="lgo", N=256, B=2, T=15,
         ess_threshold=0.9, seed=117, dtypes=("f64", "f32")),
    dict(name="lorenz_s_apf_boot", model="lorenz_s", filter="apf", proposal="bootstrap", N=200, B=3, T=15,
         ess_threshold=0.9, seed=118, nan_steps=(6,), dtypes=("f64",)),
    # ... the same observation as a vector of length one (event_shape = Size([1])): the APF + optimal proposal
    dict(name="lorenz_o1_apf_lgo", model="lorenz_o1", filter="apf", proposal="lgo", N=128, B=2, T=15,
         ess_threshold=0.9, seed=119, dtypes=("f64", "f32")),
    # D = 2 / scalar observation: y = x1 + 0.5 x2 + s v
    dict(name="rw2d_s_sisr_lgo", model="rw2d_s", filter="sisr", proposal="lgo", N=300, B=3, T=20,
         ess_threshold=0.6, seed=120, nan_steps=(5,), dtypes=("f64",)),
    dict(name="rw2d_o1_apf_lgo", model="rw2d_o1", filter="apf", proposal="lgo", N=256, B=1, T=20,
         ess_threshold=0.9, seed=121, nan_steps=(4, 12), dtypes=("f64", "f32")),
    dict(name="rw2d_s_sisr_boot", model="rw2d_s", filter="sisr", proposal="bootstrap", N=333, B=2, T=20,
         ess_threshold=0.9, seed=122, dtypes=("f64",)),
    # D = 3 / O = 3: dense A, offset, three noise scales - the full 3x3 algebra of the optimal proposal
    dict(name="lorenz_o3_apf_lgo", model="lorenz_o3", filter="apf", proposal="lgo", N=128, B=2, T=15,
         ess_threshold=0.9, seed=123, dtypes=("f64", "f32")),
    dict(name="lorenz_o3_sisr_lgo", model="lorenz_o3", filter="sisr", proposal="lgo", N=200, B=1, T=15,
         ess_threshold=0.7, seed=124, nan_steps=(9,), dtypes=("f64",)),
    dict(name="lorenz_o3_sisr_boot", model="lorenz_o3", filter="sisr", proposal="bootstrap", N=256, B=2, T=15,
         ess_threshold=0.9, seed=125, dtypes=("f64",)),
    # per-filter (theta on the batch dim) sigma / A / b / s rows of a VECTOR model: B = 3 distinct parameter sets
    dict(name="rw2d_theta_apf_lgo", model="rw2d_theta", filter="apf", proposal="lgo", N=256, B=3, T=20,
         ess_threshold=0.9, seed=126, nan_steps=(8,), dtypes=("f64", "f32")),
    dict(name="rw2d_theta_sisr_boot", model="rw2d_theta_b", filter="sisr", proposal="bootstrap", N=300, B=3, T=20,
         ess_threshold=0.8, seed=127, dtypes=("f64",)),
    dict(name="rw2d_theta_sisr_lgo", model="rw2d_theta", filter="sisr", proposal="lgo", N=128, B=3, T=20,
         ess_threshold=0.5, seed=128, dtypes=("f64",)),
    # observe_every_step = 3 (filters/base.py:204-210): two propagate-only moves before every weighted one, NaN rows too
    dict(name="lg1d_sisr_boot_oes3", model="lg1d", filter="sisr", proposal="bootstrap", N=500, B=2, T=12,
         ess_threshold=0.9, seed=129, observe_every_step=3, nan_steps=(4, 5), dtypes=("f64", "f32")),
    dict(name="sine_apf_lgo_oes3", model="sine", filter="apf", proposal="lgo", N=256, B=3, T=12,
         ess_threshold=0.9, seed=130, observe_every_step=3, nan_steps=(7,), dtypes=("f64", "f32")),
    dict(name="lorenz_sisr_boot_oes2", model="lorenz", filter="sisr", proposal="bootstrap", N=256, B=2, T=10,
         ess_threshold=0.5, seed=131, observe_every_step=2, dtypes=("f64",)),
    # a SCALAR state under a VECTOR observation (D = 1, O = 2): the ``hidden_is_1d`` branch of find_optimal_density with a matrix
    # observation (proposals/utils.py:243-245) - in this package the one linear-Gaussian shape that stays on the torch route
    # (filters/particle/proposals/linear.py::_ObservationUpdate).  SISR only: the reference's APF + LinearGaussianObservations
    # ``pre_weight`` multiplies an (O, 1) matrix into (N, B, 1) particles (proposals/linear.py:79-81) and cannot run it
    dict(name="lg1d_o2_sisr_lgo", model="lg1d_o2", filter="sisr", proposal="lgo", N=200, B=2, T=15,
         ess_threshold=0.9, seed=133, nan_steps=(6,), dtypes=("f64", "f32")),
    dict(name="lg1d_o2_sisr_boot", model="lg1d_o2", filter="sisr", proposal="bootstrap", N=128, B=3, T=15,
         ess_threshold=0.8, seed=134, dtypes=("f64",)),
    dict(name="sv_apf_boot_oes5", model="sv_batched", filter="apf", proposal="bootstrap", N=256, B=4, T=8,
         ess_threshold=0.9, seed=132, observe_every_step=5, dtypes=("f64",)),  # the SV notebook's own setting (:83)
]

# ---- round 6: reference runs at the sizes the column-CLUSTER kernel takes (2 049 .. 16 384 particles; pf_cluster.hpp) - kept apart
# from CASES: the suites parametrised over those pin the column / per-step routes of small filters, these pin the cluster kernel
# (``kernel_route`` "cluster" / "spread", tests/conftest.py) and the per-step route at the same sizes.  No smoothing arrays
# (the fixtures stay a few hundred KB).
CLUSTER_CASES = [
    # SISR: ess_threshold 0.5 - moves with and without resampling; the column = 4 member workgroups x 2 filters
    dict(name="lg1d_sisr_boot_n4096", model="lg1d", filter="sisr", proposal="bootstrap", N=4096, B=2, T=8,
         ess_threshold=0.5, seed=201, dtypes=("f64", "f32"), no_smooth=True),
    # the headline model at the per-rank size of BASELINE configs[4]: 8 members, one NaN row
    dict(name="sine_apf_lgo_n8192", model="sine", filter="apf", proposal="lgo", N=8192, B=1, T=6,
         ess_threshold=0.9, seed=202, nan_steps=(3,), dtypes=("f64", "f32"), no_smooth=True),
    # D = 3 / O = 1, and a column that ends 4 particles into its third member
    dict(name="lorenz_o1_apf_lgo_n2052", model="lorenz_o1", filter="apf", proposal="lgo", N=2052, B=2, T=6,
         ess_threshold=0.9, seed=203, dtypes=("f64", "f32"), no_smooth=True),
]

_LORENZ_A_S = [0.8, 0.0, 0.3]
_LORENZ_A_O3 = [[0.8, 0.1, 0.0], [-0.2, 0.9, 0.05], [0.0, 0.3, 0.7]]
_RW2D_A_S = [1.0, 0.5]

CASE_BY_NAME = {c["name"]: c for c in CASES + CLUSTER_CASES}
# the cases whose model the fused kernels take (D > 1 or O == 1); ``lg1d_o2_*`` runs on the torch route (tests/test_torch_route_golden.py)
FUSED_CASES = [c for c in CASES if c["model"] != "lg1d_o2"]


def build_spec(case, dtype=torch.float64) -> M.ModelSpec:
    spec = _build_spec(case, dtype)
    spec.observe_every_step = int(case.get("observe_every_step", 1))
    return spec


def _build_spec(case, dtype=torch.float64) -> M.ModelSpec:
    m, b = case["model"], case["B"]
    t = lambda v: torch.tensor(v, dtype=dtype)  # noqa: E731

    if m == "lg1d":  # tests/filters/models.py:10-15
        return M.ModelSpec(M.HID_LINEAR, (0.0, 0.99, 0.05), 0, 1.0, (0.0, 0.05), M.OBS_LINEAR, (1.0, 0.0, 0.15), 0)
    if m == "lg1d_o2":  # the same AR(1) state seen by two sensors: y = b + a x + s v, a = (1, 0.5)
        return M.ModelSpec(M.HID_LINEAR, (0.0, 0.99, 0.05), 0, 1.0, (0.0, 0.05), M.OBS_LINEAR,
                           (t([1.0, 0.5]), t([0.0, 0.1]), t([0.15, 0.2])), 2)
    if m == "sine":  # README.md:44-67
        return M.ModelSpec(M.HID_SINE_EM, (0.0, 1.0), 0, 0.1, (0.0, 1.0), M.OBS_LINEAR, (1.0, 0.0, 0.1), 0)
    if m == "sv_batched":  # stochastic-volatility.ipynb, B distinct parameter rows + B distinct series
        k = case.get("param_step_scale", 1.0)  # spacing of the B parameter rows (1: the golden fixtures)
        kappa = t([0.05 + 0.01 * k * i for i in range(b)])
        gamma = t([1.0 + 0.1 * k * i for i in range(b)])
        sigma = t([0.10 + 0.02 * k * i for i in range(b)])
        mu = t([0.0 + 0.05 * k * i for i in range(b)])
        return M.ModelSpec(M.HID_VERHULST_EM, (kappa, gamma, sigma), 0, 0.2, (1.0, 0.1), M.OBS_SV, (mu,), 0)
    if m == "lorenz":  # lorenz.ipynb
        return M.ModelSpec(
            M.HID_LORENZ63_EM, (10.0, 28.0, 8.0 / 3.0, 1.0), 3, 0.01, (t(_LORENZ_INIT[0]), t(_LORENZ_INIT[1])),
            M.OBS_LINEAR, (t(_LORENZ_A), t([0.0]), t([math.sqrt(0.1)])), 2,
        )
    if m in ("lorenz_s", "lorenz_o1", "lorenz_o3"):  # lorenz.ipynb's process under other linear observations
        hid = (M.HID_LORENZ63_EM, (10.0, 28.0, 8.0 / 3.0, 1.0), 3, 0.01, (t(_LORENZ_INIT[0]), t(_LORENZ_INIT[1])))
        if m == "lorenz_s":  # scalar observation: event_shape = Size([]), a of shape (D,)
            return M.ModelSpec(*hid, M.OBS_LINEAR, (t(_LORENZ_A_S), t(0.0), t(math.sqrt(0.1))), 0)
        if m == "lorenz_o1":  # the same observation as a vector of length one
            return M.ModelSpec(*hid, M.OBS_LINEAR, (t([_LORENZ_A_S]), t([0.0]), t([math.sqrt(0.1)])), 1)
        return M.ModelSpec(*hid, M.OBS_LINEAR, (t(_LORENZ_A_O3), t([0.1, -0.2, 0.3]), t([0.3, 0.4, 0.5])), 3)
    if m in ("rw2d_s", "rw2d_o1"):
        sig = t([0.05, 0.1])
        hid = (M.HID_LINEAR, (torch.zeros_like(sig), torch.ones_like(sig), sig), 2, 1.0, (torch.zeros_like(sig), sig))
        if m == "rw2d_s":
            return M.ModelSpec(*hid, M.OBS_LINEAR, (t(_RW2D_A_S), t(0.05), t(0.15)), 0)
        return M.ModelSpec(*hid, M.OBS_LINEAR, (t([_RW2D_A_S]), t([0.05]), t([0.15])), 1)
    if m in ("rw2d_theta", "rw2d_theta_b"):
        # B distinct (sigma, A, s) rows: theta on the batch dim of a vector model.  The offset b is shared by the filters
        # under the optimal proposal - the reference's find_optimal_density takes ``y - b`` of shape (B, O) for a MATRIX
        # (proposals/utils.py:260 ``o_inv_cov.matmul(y)``) and raises - and per filter in the ``_b`` variant (Bootstrap)
        sig = torch.stack([t([0.05 + 0.02 * (i % 4), 0.1 - 0.02 * (i % 4)]) for i in range(b)])   # (B, 2); (% 4: scales stay positive for any B)
        a = torch.stack([(1.0 + 0.25 * i) * t([[1.0, 0.2 * i], [-0.1 * i, 1.0]]) for i in range(b)])  # (B, 2, 2)
        off = torch.stack([t([0.02 * i, -0.03 * i]) for i in range(b)]) if m == "rw2d_theta_b" else t([0.02, -0.03])
        s = torch.stack([t([0.15 + 0.05 * (i % 4), 0.2 - 0.03 * (i % 4)]) for i in range(b)])      # (B, 2)
        return M.ModelSpec(
            M.HID_LINEAR, (torch.zeros_like(sig), torch.ones_like(sig), sig), 2, 1.0, (torch.zeros_like(sig), sig),
            M.OBS_LINEAR, (a, off, s), 2,
        )
    if m == "rw_rand":
        # development sweeps (tools/fuzz_parity.py, tests): a D-dimensional random walk (case["D"] in {2, 3}) under a DENSE random
        # linear observation - case["O"] in {0 (scalar, event_shape = Size([])), 1, 2, 3} - drawn from the case's seed; with
        # case["per_filter"] every filter has its own (sigma, A, s) rows (and its own offset when the proposal is Bootstrap)
        d, o = int(case["D"]), int(case["O"])
        gen = torch.Generator().manual_seed(int(case["seed"]) * 31 + 7)
        rows = (b,) if case.get("per_filter") else ()
        od = max(o, 1)
        sig = (0.05 + 0.1 * torch.rand(rows + (d,), generator=gen, dtype=torch.float64)).to(dtype)
        a = 0.5 * torch.randn(rows + (od, d), generator=gen, dtype=torch.float64) + (torch.eye(od, d, dtype=torch.float64) if od <= d else 0.0)
        off_rows = rows if case.get("proposal") == "bootstrap" else ()  # (the reference's optimal proposal cannot take a per-filter offset)
        off = 0.2 * torch.randn(off_rows + (od,), generator=gen, dtype=torch.float64)
        sc = 0.1 + 0.2 * torch.rand(rows + (od,), generator=gen, dtype=torch.float64)
        a, off, sc = a.to(dtype), off.to(dtype), sc.to(dtype)
        if o == 0:
            a, off, sc = a[..., 0, :], off[..., 0], sc[..., 0]
        return M.ModelSpec(M.HID_LINEAR, (torch.zeros_like(sig), torch.ones_like(sig), sig), d, 1.0, (torch.zeros_like(sig), sig),
                           M.OBS_LINEAR, (a, off, sc), o)
    if m == "rw2d":  # tests/filters/models.py:28-52: x' = I2 x + (0.05, 0.1) e, x0 ~ N(0, sigma), y ~ N(I2 x, 0.15)
        sig = t([0.05, 0.1])
        return M.ModelSpec(
            M.HID_LINEAR, (torch.zeros_like(sig), torch.ones_like(sig), sig), 2, 1.0, (torch.zeros_like(sig), sig),
            M.OBS_LINEAR, (torch.eye(2, dtype=dtype), t([0.0, 0.0]), t([0.15, 0.15])), 2,
        )
    if m == "ou_batched":  # tests/inference/models.py:12-33 with theta on the batch dim
        kappa = t([0.025 * (i + 1) for i in range(b)])
        gamma = t([0.0 + 0.1 * i for i in range(b)])
        sigma = t([0.05 + 0.01 * i for i in range(b)])
        return M.ModelSpec(M.HID_OU, (kappa, gamma, sigma), 0, 1.0, (0.0, 0.1), M.OBS_LINEAR, (1.0, 0.0, 0.05), 0)
    raise KeyError(m)


def simulate(case, spec: M.ModelSpec, dtype=torch.float64) -> torch.Tensor:
    """Synthetic observations from *our own* simulator of the same model (first batch column's parameters when the
    model is theta-batched, except the SV case which simulates B distinct series -> ``y (T,B)``)."""
    g = torch.Generator().manual_seed(case["seed"] + 7)
    t_len = case["T"]
    b = case["B"]
    per_series = case["model"] == "sv_batched"
    shape = (1, b) if (per_series or (case["model"] in ("ou_batched", "rw2d_theta", "rw2d_theta_b") or bool(case.get("per_filter")))) else (1, 1)
    if spec.dim > 0:
        shape = shape + (spec.dim,)
    x = M.initial_sample(spec, torch.randn(shape, generator=g, dtype=dtype))
    ys = []
    for _ in range(t_len):
        for _ in range(spec.observe_every_step):  # (observed at every observe_every_step-th move of the process)
            x = M.propagate(spec, x, torch.randn(shape, generator=g, dtype=dtype))
        loc, scale = M.obs_loc_scale(spec, x)
        scale = M._t(scale, loc)
        yv = loc + scale * torch.randn(loc.shape, generator=g, dtype=dtype)
        ys.append(yv[0] if per_series else yv[0, 0])
    y = torch.stack(ys)
    for s in case.get("nan_steps", ()):
        y[s] = float("nan")
    return y
