"""
TEST INFRASTRUCTURE — generates ``tests/golden/*.npz`` by running the **unmodified reference**
(``/root/reference/pyfilter``, imported behind ``oracle/ref_shim``) in THIS container.

    python oracle/make_golden.py            # regenerates every fixture (fp64 and fp32 runs, one subprocess each)

What is recorded per case: inputs ``y, x0, z tape (T,N,B,[D]) , u tape (T,B)``, model spec, and the
reference's outputs: per-step ``x, w, ll, idx``, and ``filter_means, filter_variance, loglikelihood``; the reference's
``smooth(states, "fl")`` over all T + 1 states (deterministic given the states); for the cases in ``FFBS_CASES`` the
mean / variance over trajectories of ``smooth(states, "ffbs")`` (its ``Categorical`` draws are not injectable:
statistical fixture); for the cases in ``STATE_DICT_CASES`` the reference's ``FilterResult.state_dict()`` after
``STATE_DICT_AT`` observations, flattened to ``sd::<path>`` keys (checkpoint interoperability).

Tape injection (the reference's arithmetic is untouched):
* ``torch.normal(mean, std)`` is wrapped to draw ``z = randn(shape)`` (float32, upcast), record it and return
  ``z*std + mean`` - the same two aten ops the real ``normal_out`` issues (``normal_(0,1); mul_(std); add_(mean)``);
  ``MultivariateNormal``'s ``_standard_normal`` is wrapped the same way.
* the ``resampling=`` ctor argument receives a callable that draws ``u`` (float32), records it and calls the
  reference's ``systematic(w, normalized=..., u=u)``.  ``batch_shape=[B>=1]`` always: the reference drops ``u``
  for 1-D weights (resampling.py:14).
* ``observe_every_step > 1`` (round 5): one ``filter()`` call is several moves (filters/base.py:204-210); the tapes are then
  PER MOVE - ``u_tape (moves, B)`` drawn at the top of every ``predict`` (the only place a SISR resamples; the APF's
  resampling in ``correct`` follows the call's last ``predict``), ``z_tape (moves, N, B, [D])`` one lazily drawn transition
  per move, in order - and ``move_of_obs (T,)`` names the move that consumed each observation.

    python oracle/make_golden.py --only lorenz_s_sisr_lgo,...   # (re)generates the named cases only

The fixtures are data only.  The reference source never leaves this container.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(ROOT, "tests", "golden")
FFBS_CASES = ("lg1d_sisr_boot", "sine_apf_lgo", "lorenz_sisr_boot", "sv_apf_boot", "rw2d_sisr_boot")
STATE_DICT_CASES = ("lg1d_apf_lgo", "lorenz_sisr_boot", "sv_sisr_boot", "rw2d_apf_lgo")
STATE_DICT_AT = 12  # observations consumed when the checkpoint is taken


def _main_child(dtype_name: str, only=None):
    import math

    import numpy as np
    import torch

    dtype = {"f64": torch.float64, "f32": torch.float32}[dtype_name]
    torch.set_default_dtype(dtype)

    sys.path.insert(0, os.path.join(HERE, "ref_shim"))
    sys.path.insert(1, "/root/reference")
    sys.path.insert(2, ROOT)

    import pyfilter  # noqa: F401  (the reference)
    from pyfilter.filters.particle import APF, SISR, proposals
    from pyfilter.resampling import systematic as ref_systematic, multinomial as ref_multinomial  # noqa: F401
    from pyfilter.utils import get_ess as ref_get_ess, normalize as ref_normalize
    from pyfilter.filters.particle.utils import log_likelihood as ref_ll
    from stochproc import timeseries as ts
    from torch.distributions import Independent, Normal
    import torch.distributions.multivariate_normal as mvn_mod

    from oracle import models as M
    from oracle.cases import CASES, CLUSTER_CASES, build_spec, simulate

    # ---------------------------------------------------------------------------------------------------------
    class Tape:
        def __init__(self):
            self.z, self.u, self.mask = [], [], None
            self.cur_u = None

    tape = Tape()
    real_normal = torch.normal

    def taped_normal(mean, std, *args, **kwargs):
        if not (isinstance(mean, torch.Tensor) and isinstance(std, torch.Tensor)):
            return real_normal(mean, std, *args, **kwargs)
        z32 = torch.randn(mean.shape, dtype=torch.float32)
        tape.z.append(z32)
        return z32.to(mean.dtype) * std + mean

    def taped_standard_normal(shape, dtype, device):
        z32 = torch.randn(shape, dtype=torch.float32)
        tape.z.append(z32)
        return z32.to(dtype)

    torch.normal = taped_normal
    mvn_mod._standard_normal = taped_standard_normal

    # ---------------------------------------------------------------------------------------------------------
    def build_reference_model(spec: M.ModelSpec):
        """Reference-side model objects (shim classes) for a ModelSpec; definitions follow README.md:44-67,
        examples/lorenz.ipynb, examples/stochastic-volatility.ipynb, tests/filters/models.py."""
        k = spec.hidden
        hp = tuple(torch.as_tensor(p, dtype=dtype) for p in spec.hidden_params)
        m0, s0 = (torch.as_tensor(v, dtype=dtype) for v in spec.init)
        d = spec.dim

        def init_kernel(*_):
            n = Normal(m0, s0)
            return Independent(n, 1) if d > 0 else n

        inc = Normal(torch.tensor(0.0), torch.tensor(spec.inc_scale))
        if d > 0:
            inc = Independent(inc.expand(torch.Size([d])), 1)

        if k == M.HID_LINEAR and d > 0:
            # the reference's own construction of its 2-D model (tests/filters/models.py:31-38): LinearModel((A, sigma), ...)
            # with A the identity - the diagonal walk the product's RandomWalk(dim=2) restates
            assert all(bool((q == v).all()) for q, v in zip(hp[:2], (0.0, 1.0)))
            hidden = ts.LinearModel((torch.eye(d, dtype=dtype), hp[2]), inc, init_kernel)
        elif k == M.HID_LINEAR:
            hidden = ts.AffineProcess(lambda x, a, b, s: (a + b * x.value, s), hp, inc, init_kernel)
        elif k == M.HID_SINE_EM:
            hidden = ts.AffineEulerMaruyama(
                lambda x, g, s: (torch.sin(x.value - g), s), hp, inc, spec.dt, init_kernel
            )
        elif k == M.HID_VERHULST_EM:
            hidden = ts.AffineEulerMaruyama(
                lambda x, ka, g, s: (ka * (g - x.value) * x.value, s * x.value), hp, inc, spec.dt, init_kernel
            )
        elif k == M.HID_LORENZ63_EM:

            def f(x, s, r, b, sigma):
                v = x.value
                x_t = -s * (v[..., 0] - v[..., 1])
                y_t = r * v[..., 0] - v[..., 1] - v[..., 0] * v[..., 2]
                z_t = v[..., 0] * v[..., 1] - b * v[..., 2]
                return torch.stack((x_t, y_t, z_t), dim=-1), sigma

            hidden = ts.AffineEulerMaruyama(f, hp, inc, spec.dt, init_kernel)
        elif k == M.HID_OU:
            dt = spec.dt

            def ou(x, ka, g, s):
                e = torch.exp(-ka * dt)
                return g + (x.value - g) * e, s * torch.sqrt((1.0 - torch.exp(-2.0 * ka * dt)) / (2.0 * ka))

            hidden = ts.AffineProcess(ou, hp, inc, init_kernel)
        else:
            raise NotImplementedError(k)

        op = tuple(torch.as_tensor(p, dtype=dtype) for p in spec.obs_params)
        if spec.obs == M.OBS_LINEAR:
            es = torch.Size([spec.obs_dim]) if spec.obs_dim > 0 else torch.Size([])
            return ts.LinearStateSpaceModel(hidden, op, es, observe_every_step=spec.observe_every_step)
        if spec.obs == M.OBS_SV:
            return ts.StateSpaceModel(hidden, lambda x, mu: Normal(mu, x.value), op, observe_every_step=spec.observe_every_step)
        raise NotImplementedError(spec.obs)

    def flatten_state_dict(sd, prefix="sd"):
        """Nested dict of tensors -> ``{"sd::a::b": ndarray}`` (copies: the checkpoint is a snapshot)."""
        flat = {}
        for k, v in sd.items():
            key = f"{prefix}::{k}"
            if isinstance(v, dict):
                flat.update(flatten_state_dict(v, key))
            else:
                flat[key] = torch.as_tensor(v).detach().clone().numpy()
        return flat

    # ---------------------------------------------------------------------------------------------------------
    os.makedirs(GOLDEN, exist_ok=True)

    for case in CASES + CLUSTER_CASES:
        if dtype_name not in case["dtypes"] or (only is not None and case["name"] not in only):
            continue
        oes = int(case.get("observe_every_step", 1))
        torch.manual_seed(case["seed"])
        spec = build_spec(case, dtype)
        ssm = build_reference_model(spec)
        n, b, t_len = case["N"], case["B"], case["T"]

        y = simulate(case, spec, dtype)

        filt_cls = {"sisr": SISR, "apf": APF}[case["filter"]]

        class Taped(filt_cls):
            def predict(self, state):
                if oes > 1:  # per-move tapes: a fresh u at the top of every predict
                    tape.cur_u = torch.rand(b, dtype=torch.float32)
                    u_tape.append(tape.cur_u)
                if case["filter"] == "sisr":
                    w_ = ref_normalize(state.weights.clone())
                    tape.mask = ref_get_ess(w_, normalized=True) < self._resample_threshold
                else:
                    tape.mask = torch.ones(b, dtype=torch.bool)
                return super().predict(state)

        def taped_resampler(w, normalized=False):
            u = tape.cur_u[tape.mask].reshape(-1, 1).to(w.dtype)
            assert w.dim() == 2 and w.shape[1] == u.shape[0]
            return ref_systematic(w, normalized=normalized, u=u)

        prop = {"bootstrap": proposals.Bootstrap, "lgo": proposals.LinearGaussianObservations}[case["proposal"]]()
        filt = Taped(ssm, n, resampling=taped_resampler, proposal=prop, ess_threshold=case["ess_threshold"])
        filt.set_batch_shape(torch.Size([b]))

        tape.z.clear()
        state = filt.initialize()
        z0 = tape.z.pop()
        assert not tape.z
        x0 = state.timeseries_state.value.clone()
        result = filt.initialize_with_result(state)

        steps = {k: [] for k in ("x", "w", "ll", "idx")}
        u_tape, z_tape, move_of_obs = [], [], []
        all_states, checkpoint = [state], None
        for t in range(t_len):
            if t == STATE_DICT_AT and case["name"] in STATE_DICT_CASES:
                checkpoint = flatten_state_dict(result.state_dict())
            if oes == 1:
                tape.cur_u = torch.rand(b, dtype=torch.float32)
                u_tape.append(tape.cur_u)
            t_before = int(state.timeseries_state.time_index)
            state = filt.filter(y[t], state, result=result)
            _ = state.timeseries_state.value  # force the lazy sample
            moves = int(state.timeseries_state.time_index) - t_before
            assert len(tape.z) == moves and (oes > 1 or moves == 1), (len(tape.z), moves)
            z_tape.extend(tape.z)
            tape.z.clear()
            move_of_obs.append(len(z_tape) - 1)
            steps["x"].append(state.timeseries_state.value.clone())
            steps["w"].append(state.weights.clone())
            steps["ll"].append(state.get_loglikelihood().clone())
            steps["idx"].append(state.previous_indices.clone())
            all_states.append(state)

        out = {
            "y": y.numpy(),
            "x0": x0.numpy(),
            "z0": z0.numpy(),
            "z_tape": torch.stack(z_tape).numpy(),
            "u_tape": torch.stack(u_tape).numpy(),
            "filter_means": result.filter_means.numpy(),
            "filter_variance": result.filter_variance.numpy(),
            "loglikelihood": result.loglikelihood.numpy(),
        }
        if oes > 1:
            assert len(u_tape) == len(z_tape)
            out["move_of_obs"] = np.asarray(move_of_obs, dtype=np.int64)
        for k, v in steps.items():
            out[f"step_{k}"] = torch.stack(v).numpy()
        # smoothing over the recorded states (particle/base.py:105-157); everything the filtering part of the fixture
        # holds was produced above - the calls below only consume further random numbers
        if oes == 1 and not case.get("no_smooth"):  # (recorded states of a thinned run skip moves: their ancestors do not chain)
            out["smooth_fl"] = filt.smooth(all_states, "fl").numpy()
        if case["name"] in FFBS_CASES and dtype_name == "f64":
            tape.mask = torch.ones(b, dtype=torch.bool)
            tape.cur_u = torch.rand(b, dtype=torch.float32)
            draws = [filt.smooth(all_states, "ffbs") for _ in range(4)]  # 4 independent backward passes
            traj = torch.stack(draws)  # (4, T + 1, N, B, [D])
            out["ffbs_u_last"] = tape.cur_u.numpy()
            out["ffbs_mean"] = traj.mean(dim=(0, 2)).numpy()
            out["ffbs_var"] = traj.var(dim=(0, 2)).numpy()
        if checkpoint is not None:
            out.update(checkpoint)
        path = os.path.join(GOLDEN, f"{case['name']}_{dtype_name}.npz")
        np.savez_compressed(path, **out)
        print(f"wrote {path}: ll={result.loglikelihood.tolist()}")

    # ---------------------------------------------------------------------------------------------------------
    if only is not None:
        return
    # primitives: the reference's own known-answer arrangement for systematic() (tests/test_resampling.py:31-47)
    # plus correctly-normalised rows, normalize()/get_ess()/log_likelihood() edge cases.
    # ---------------------------------------------------------------------------------------------------------
    torch.manual_seed(123)
    prim = {}
    w_t = ref_normalize(torch.randn((10, 300), dtype=torch.float64))  # as in the reference's test (axis quirk kept)
    u_t = torch.rand(w_t.shape, dtype=torch.float64)
    prim["ka_w"] = w_t.numpy()
    prim["ka_u"] = u_t.numpy()
    prim["ka_idx"] = ref_systematic(w_t.moveaxis(0, 1), u=u_t, normalized=True).moveaxis(0, 1).numpy()

    for nm, (n_, b_) in {"a": (257, 5), "b": (4096, 3), "c": (1000, 1)}.items():
        lw = (3.0 * torch.randn(n_, b_)).to(dtype)
        if nm == "a":
            lw[3, 0] = float("nan")
            lw[7, 1] = float("inf")
            lw[:, 2] = -float("inf")
            lw[::2, 3] = -float("inf")
        prim[f"norm_{nm}_in"] = lw.clone().numpy()
        W = ref_normalize(lw)  # in place: lw now holds the nan_to_num'ed values
        prim[f"norm_{nm}_inplace"] = lw.numpy()
        prim[f"norm_{nm}_W"] = W.numpy()
        prim[f"norm_{nm}_ess"] = ref_get_ess(W, normalized=True).numpy()
        u_ = torch.rand(b_, 1).to(dtype)
        prim[f"norm_{nm}_u"] = u_.numpy()
        prim[f"norm_{nm}_idx"] = ref_systematic(W, normalized=True, u=u_).numpy()
        v = torch.randn(n_, b_).to(dtype)
        prim[f"norm_{nm}_v"] = v.numpy()
        prim[f"norm_{nm}_ll_w"] = ref_ll(v, W).numpy()
        prim[f"norm_{nm}_ll"] = ref_ll(v).numpy()
    np.savez_compressed(os.path.join(GOLDEN, f"primitives_{dtype_name}.npz"), **prim)
    print("wrote primitives", dtype_name)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        _main_child(sys.argv[2], set(sys.argv[3].split(",")) if len(sys.argv) > 3 else None)
    else:
        extra = [sys.argv[2]] if len(sys.argv) > 2 and sys.argv[1] == "--only" else []
        for dt in ("f64", "f32"):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", dt] + extra, cwd=ROOT)
