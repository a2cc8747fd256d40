"""
TEST INFRASTRUCTURE - generates ``tests/golden/inference_*.npz``: event logs of the **unmodified reference's** SMC^2 /
PMMH code (``/root/reference/pyfilter/inference``, imported behind ``oracle/ref_shim``) running in THIS container -
SURVEY.md section 8(f) row 2: ``sequential/smc2.py:53-65``, ``sequential/state.py:35-44``, ``sequential/kernels/mh.py:52-140``
(``ParticleMetropolisHastings.update`` incl. ``_increase_states``), ``batch/mcmc/utils.py:14-77`` (``run_pmmh``),
``inference/utils.py:42-76`` (``construct_mvn``), ``batch/mcmc/proposals/{symmetric_mh,random_walk}.py``.

    python oracle/make_golden_inference.py          # regenerates every inference fixture (float64)

A fixture is an ordered list of EVENTS, each holding the random numbers the reference consumed at that point and what it
computed from them, flattened to ``e<k>::<kind>::<field>`` arrays:

    theta0        the theta-particles the run starts from (constrained), the observations
    init          the filters' initial sample: z0 (N, B)
    move          one online ``filter.filter(y_t, state)``: u (B,), z (N, B) -> ll_t (B,), the theta-weights and ESS after
                  ``state.append`` (sequential/state.py:35-44)
    rejuvenate    ``ParticleMetropolisHastings.update`` starts: the resampling uniform -> ancestors of the theta-particles,
                  the Gaussian proposal fitted BEFORE the resampling (mean, scale_tril)
    pmmh_draw     ``run_pmmh``: the proposal's standard normals eps (B, P) -> theta* (unconstrained)
    run           a whole ``batch_filter`` over the parsed data (the proposal filter of a PMMH move, or the re-run of
                  ``_increase_states``): z0, z (t, N, B), u (t, B) -> loglikelihood (B,)
    pmmh_accept   the reference's own ``log_acc_prob`` (B,), the acceptance uniforms, the accepted mask, and theta /
                  log-likelihoods / last-state moments after the exchange
    rejuvenated   how the update ended ("done" / "increase"), theta, theta-weights, log-likelihoods, particle count
    final         filter means / variances of the whole run, log-likelihoods, weights, ESS history

Tape injection leaves the reference's arithmetic untouched (same technique as ``make_golden.py``): ``torch.normal`` and
``MultivariateNormal``'s ``_standard_normal`` draw float32 standard normals that are recorded; the filters get a
resampler that draws and records ``u`` and calls the reference's ``systematic(..., u=u)``; the theta-level resampler does
the same through a (B, 1) view (the reference drops ``u`` for 1-D weights, resampling.py:14); the acceptance uniforms are
recorded by wrapping ``Tensor.uniform_``; ``log_acc_prob`` is captured through the ``torch`` name of
``batch/mcmc/utils.py`` (a forwarding proxy whose ``empty_like`` remembers its argument).

The model is the Ornstein-Uhlenbeck state-space model of ``tests/inference/models.py:12-33`` with its priors
(``stochproc.models.OrnsteinUhlenbeck`` is not in the tree - row M, parity unpinned at that boundary - so the process is
restated here exactly as in ``make_golden.py``: exact discretisation, stationary initial distribution).

The fixtures are data only.  The reference source never leaves this container.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(ROOT, "tests", "golden")

# name: B theta-particles, N state particles, T observations, SMC2 kwargs
CASES = {
    # two PMMH moves per rejuvenation, several rejuvenations, no particle doubling
    "inference_smc2_ou": dict(B=12, N=96, T=18, seed=11, threshold=0.5, kwargs=dict(num_steps=2)),
    # the adaptive stopping rule (distance_threshold, kernels/mh.py:92-100)
    "inference_smc2_ou_adaptive": dict(B=10, N=64, T=14, seed=5, threshold=0.6, kwargs=dict(num_steps=6, distance_threshold=0.5)),
    # so few state particles / so high an acceptance bar that the kernel doubles the particles (_increase_states)
    "inference_smc2_ou_increase": dict(B=12, N=16, T=12, seed=3, threshold=0.6, kwargs=dict(num_steps=2, acceptance_threshold=0.6)),
    # PMMH proper: parallel chains, the random-walk kernel re-centred after accepted moves (mutate_kernel=True)
    "inference_pmmh_ou_rw": dict(B=6, N=64, T=12, seed=7, pmmh=dict(num_samples=5, scale=0.05)),
}


def main():
    import numpy as np
    import torch

    torch.set_default_dtype(torch.float64)
    sys.path.insert(0, os.path.join(HERE, "ref_shim"))
    sys.path.insert(1, "/root/reference")
    sys.path.insert(2, ROOT)

    import pyfilter  # noqa: F401  (the reference)
    import torch.distributions.multivariate_normal as mvn_mod
    from pyfilter import inference as inf
    from pyfilter.filters.particle import APF, proposals
    from pyfilter.inference.batch.mcmc import utils as mcmc_utils
    from pyfilter.inference.batch.mcmc.proposals import RandomWalk, SymmetricMH
    from pyfilter.inference.sequential.kernels import mh as mh_mod
    from pyfilter.resampling import systematic as ref_systematic
    from pyro.distributions import Exponential, LogNormal, Normal
    from stochproc import timeseries as ts

    # ---------------------------------------------------------------------------------------------------------------
    class Recorder:
        def __init__(self):
            self.events = []
            self.sinks = []      # stack of lists collecting the standard normals drawn through torch.normal
            self.run = None      # the batch_filter run in progress
            self.cur_u = None
            self.eps = []        # theta-level standard normals (MultivariateNormal / Normal.sample of the proposal)
            self.log_acc = []
            self.u_acc = []
            self.kernels = []
            self.theta_mode = False  # inside run_pmmh's proposal draw: torch.normal belongs to the theta level

        def emit(self, kind, **fields):
            self.events.append((kind, {k: (v.detach().clone() if isinstance(v, torch.Tensor) else v) for k, v in fields.items()}))

    rec = Recorder()
    real_normal = torch.normal

    def taped_normal(mean, std, *args, **kwargs):
        if not (isinstance(mean, torch.Tensor) and isinstance(std, torch.Tensor)):
            return real_normal(mean, std, *args, **kwargs)
        z32 = torch.randn(mean.shape, dtype=torch.float32)
        if rec.theta_mode:
            rec.eps.append(z32.double())
        elif rec.sinks:
            rec.sinks[-1].append(z32)
        return z32.to(mean.dtype) * std + mean

    def taped_standard_normal(shape, dtype, device):
        z32 = torch.randn(shape, dtype=torch.float32)
        rec.eps.append(z32.double())
        return z32.to(dtype)

    torch.normal = taped_normal
    mvn_mod._standard_normal = taped_standard_normal

    real_uniform_ = torch.Tensor.uniform_

    def taped_uniform_(self, *a, **k):
        out = real_uniform_(self, *a, **k)
        rec.u_acc.append(out.detach().clone())
        return out

    class TorchProxy:
        """Stands in for the name ``torch`` inside batch/mcmc/utils.py: forwards everything, remembers the argument of
        ``empty_like`` - which is the reference's ``log_acc_prob`` (utils.py:69)."""

        def __getattr__(self, name):
            return getattr(torch, name)

        @staticmethod
        def empty_like(t, *a, **k):
            rec.log_acc.append(t.detach().clone())
            return torch.empty_like(t, *a, **k)

    mcmc_utils.torch = TorchProxy()

    # ---------------------------------------------------------------------------------------------------------------
    def filter_resampler(w, normalized=False):
        u = rec.cur_u.reshape(-1, 1).to(w.dtype)
        assert w.dim() == 2 and w.shape[1] == u.shape[0]
        return ref_systematic(w, normalized=normalized, u=u)

    def theta_resampler(w, normalized=False):
        assert w.dim() == 1
        u = torch.rand((), dtype=torch.float64)
        idx = ref_systematic(w.unsqueeze(1), normalized=normalized, u=u.reshape(1, 1).to(w.dtype)).squeeze(1)
        rec.emit("rejuvenate", u=u, W=w, indices=idx)  # (the proposal built right after it is attached by `update` below)
        return idx

    class TapedAPF(APF):
        def initialize(self):
            rec.sinks.append([])
            st = super().initialize()
            zs = rec.sinks.pop()
            assert len(zs) == 1, len(zs)
            if rec.run is not None:
                rec.run["z0"] = zs[0]
            else:
                rec.emit("init", z0=zs[0])
            return st

        def filter(self, y, state, result=None):
            b = self.batch_shape[0]
            rec.cur_u = torch.rand(b, dtype=torch.float32).double()
            rec.sinks.append([])
            new = super().filter(y, state, result=result)
            _ = new.timeseries_state.value  # force the lazy sample
            zs = rec.sinks.pop()
            assert len(zs) == 1, len(zs)
            if rec.run is not None:
                rec.run["z"].append(zs[0])
                rec.run["u"].append(rec.cur_u)
            else:
                # (clones: a rejuvenation triggered by this very move resamples / exchanges the state's tensors in place)
                rec.pending_move = dict(y=y.clone(), z=zs[0], u=rec.cur_u, ll=new.get_loglikelihood().clone())
            return new

        def batch_filter(self, y, bar=True, init_state=None):
            assert rec.run is None
            rec.run = dict(z=[], u=[])
            res = super().batch_filter(y, bar=False, init_state=init_state)
            run, rec.run = rec.run, None
            rec.emit("run", z0=run["z0"], z=torch.stack(run["z"]), u=torch.stack(run["u"]), ll=res.loglikelihood,
                     n=int(self._base_particles[0]), t=len(run["z"]))
            return res

    class TapedSymmetricMH(SymmetricMH):
        def build(self, context, state, filter_, y):
            d = super().build(context, state, filter_, y)
            rec.kernels.append((d.loc.clone(), d.scale_tril.clone()))
            return d

    class TapedRandomWalk(RandomWalk):
        def build(self, context, state, filter_, y):
            d = super().build(context, state, filter_, y)
            rec.kernels.append((d.mean.clone(), d.stddev.clone()))
            return d

    real_run_pmmh = mcmc_utils.run_pmmh

    def taped_run_pmmh(context, state, proposal, proposal_kernel, proposal_filter, proposal_context, y,
                       size=torch.Size([]), mutate_kernel=False):
        n_eps, n_acc, n_u, n_k = len(rec.eps), len(rec.log_acc), len(rec.u_acc), len(rec.kernels)
        real_sample = proposal_kernel.sample

        def sample(shape=torch.Size()):
            rec.theta_mode = True  # (the random walk's Normal.sample goes through torch.normal)
            try:
                rvs = real_sample(shape)
            finally:
                rec.theta_mode = False
            assert len(rec.eps) == n_eps + 1
            rec.emit("pmmh_draw", eps=rec.eps[-1].reshape(rvs.shape), rvs=rvs,
                     kernel_loc=proposal_kernel.mean, kernel_scale=getattr(proposal_kernel, "scale_tril", proposal_kernel.stddev))
            return rvs

        proposal_kernel.sample = sample
        torch.Tensor.uniform_ = taped_uniform_
        try:
            accepted = real_run_pmmh(context, state, proposal, proposal_kernel, proposal_filter, proposal_context, y, size,
                                     mutate_kernel=mutate_kernel)
        finally:
            torch.Tensor.uniform_ = real_uniform_
            del proposal_kernel.sample
        assert len(rec.log_acc) == n_acc + 1 and len(rec.u_acc) == n_u + 1 and len(rec.kernels) == n_k + 1
        latest = state.filter_state.latest_state
        rec.emit("pmmh_accept", log_acc=rec.log_acc[-1], u=rec.u_acc[-1], accepted=accepted,
                 new_kernel_loc=rec.kernels[-1][0], new_kernel_scale=rec.kernels[-1][1],
                 theta=context.stack_parameters(constrained=True), ll=state.filter_state.loglikelihood,
                 last_mean=latest.get_mean(), last_var=latest.get_variance(),
                 kernel_loc_after=proposal_kernel.mean, kernel_scale_after=getattr(proposal_kernel, "scale_tril", proposal_kernel.stddev))
        return accepted

    mh_mod.run_pmmh = taped_run_pmmh

    # ---------------------------------------------------------------------------------------------------------------
    def ou(kappa, gamma, sigma, dt=1.0):
        def ms(x, k, g, s):
            e = torch.exp(-k * dt)
            return g + (x.value - g) * e, s * torch.sqrt((1.0 - torch.exp(-2.0 * k * dt)) / (2.0 * k))

        inc = torch.distributions.Normal(torch.tensor(0.0), torch.tensor(1.0))
        return ts.AffineProcess(ms, (kappa, gamma, sigma), inc, lambda k, g, s: torch.distributions.Normal(g, s / torch.sqrt(2.0 * k)))

    def build_model(cntxt):  # tests/inference/models.py:22-33
        kappa = cntxt.named_parameter("kappa", Exponential(rate=10.0))
        gamma = cntxt.named_parameter("gamma", Normal(loc=0.0, scale=1.0))
        sigma = cntxt.named_parameter("sigma", LogNormal(loc=-2.0, scale=1.0))
        return ts.LinearStateSpaceModel(ou(kappa, gamma, sigma), (torch.tensor(1.0), torch.tensor(0.05)), torch.Size([]))

    def simulate(t_len, seed):  # OU(0.025, 0, 0.05) observed with noise 0.05 (models.py:13-19)
        import math

        g = torch.Generator().manual_seed(seed)
        x, ys = 0.0, []
        for _ in range(t_len):
            x = x * math.exp(-0.025) + 0.05 * math.sqrt((1 - math.exp(-0.05)) / 0.05) * torch.randn((), generator=g).item()
            ys.append(x + 0.05 * torch.randn((), generator=g).item())
        return torch.tensor(ys, dtype=torch.float64)

    def flatten(events):
        out = {}
        for k, (kind, fields) in enumerate(events):
            for name, v in fields.items():
                out[f"e{k:04d}::{kind}::{name}"] = np.asarray(v.numpy() if isinstance(v, torch.Tensor) else v)
            if not fields:
                out[f"e{k:04d}::{kind}::_"] = np.zeros(0)
        return out

    os.makedirs(GOLDEN, exist_ok=True)
    for name, case in CASES.items():
        torch.manual_seed(case["seed"])
        rec.__init__()
        y = simulate(case["T"], 100 + case["seed"])
        b, n = case["B"], case["N"]

        if "pmmh" not in case:
            with inf.make_context() as context:
                filt = TapedAPF(build_model, n, resampling=filter_resampler, proposal=proposals.LinearGaussianObservations())
                alg = inf.sequential.SMC2(filt, b, threshold=case["threshold"], kernel=TapedSymmetricMH(),
                                          resampling=theta_resampler, **case["kwargs"])
                kernel = alg._kernel
                real_update, real_increase = kernel.update, kernel._increase_states

                def update(context_, filter_, state_):
                    n_before = len(rec.events)
                    out = real_update(context_, filter_, state_)
                    # the proposal is built right after the resampling draw (mh.py:53-54): attach it to that event
                    k_ev = next(i for i in range(n_before, len(rec.events)) if rec.events[i][0] == "rejuvenate")
                    first_kernel = rec.kernel_at_rejuvenation
                    rec.events[k_ev][1]["kernel_mean"], rec.events[k_ev][1]["kernel_scale_tril"] = first_kernel
                    rec.emit("rejuvenated", outcome=np.array("increase" if rec.increased else "done"),
                             theta=context_.stack_parameters(constrained=True), w=out.w,
                             ll=out.filter_state.loglikelihood, n=int(filter_._base_particles[0]),
                             acceptance_moves=np.array(rec.moves_in_update))
                    return out

                def counting_run_pmmh(*a, **k):
                    if rec.moves_in_update == 0:
                        rec.kernel_at_rejuvenation = rec.kernels[-1]
                    rec.moves_in_update += 1
                    return taped_run_pmmh(*a, **k)

                def increase(filter_, state_, ctx_):
                    rec.increased = True
                    return real_increase(filter_, state_, ctx_)

                def guarded_update(context_, filter_, state_):
                    rec.moves_in_update, rec.increased = 0, False
                    return update(context_, filter_, state_)

                mh_mod.run_pmmh = counting_run_pmmh
                kernel.update, kernel._increase_states = guarded_update, increase

                state = alg.initialize()
                rec.events.insert(0, ("theta0", dict(theta=context.stack_parameters(constrained=True).clone(), y=y.clone(),
                                                      names=np.array(list(context.parameters.keys())))))
                for t in range(case["T"]):
                    n_ev = len(rec.events)
                    rec.pending_move = None
                    # (the online move is emitted first, the rejuvenation it may trigger after it)
                    state = alg.step(y[t], state)
                    mv = rec.pending_move
                    ess_hist = state.ess
                    rec.events.insert(n_ev, ("move", dict(y=mv["y"].clone(), z=mv["z"].clone(), u=mv["u"].clone(), ll=mv["ll"].clone(),
                                                          ess_after=ess_hist[t + 1].clone())))
                rec.emit("final", filter_means=state.filter_state.filter_means, filter_variance=state.filter_state.filter_variance,
                         ll=state.filter_state.loglikelihood, w=state.w, ess=state.ess,
                         theta=context.stack_parameters(constrained=True), n=int(filt._base_particles[0]))
                mh_mod.run_pmmh = taped_run_pmmh
        else:
            cfg = case["pmmh"]
            with inf.make_context() as context:
                filt = TapedAPF(build_model, n, resampling=filter_resampler, proposal=proposals.LinearGaussianObservations())
                alg = inf.batch.mcmc.PMMH(filt, cfg["num_samples"], num_chains=b, proposal=TapedRandomWalk(cfg["scale"]))
                import pyfilter.inference.batch.mcmc.pmmh as pmmh_mod

                pmmh_mod.run_pmmh = taped_run_pmmh
                # distinct starting points per chain (the reference's "mean" initializer puts every chain on the same
                # point): initialise as the reference does, then spread the chains and re-filter
                st = alg.initialize(y)
                rec.events.clear()
                with torch.no_grad():
                    for p_name, p in context.parameters.items():
                        p.mul_(1.0 + 0.1 * torch.arange(b, dtype=p.dtype))
                first = filt.batch_filter(y, bar=False)
                st = inf.batch.mcmc.state.PMMHResult(dict(context.get_parameters()), first)
                rec.events.insert(0, ("theta0", dict(theta=context.stack_parameters(constrained=True).clone(), y=y.clone(),
                                                      names=np.array(list(context.parameters.keys())))))
                proposal_filter = filt.copy()
                prop_dist = alg._proposal.build(context, st, filt, y)
                rec.emit("kernel0", loc=prop_dist.mean, scale=prop_dist.stddev)
                with context.make_new() as sub_context:
                    sub_context.set_batch_shape(torch.Size([b]))
                    proposal_filter.initialize_model(sub_context)
                for _ in range(cfg["num_samples"]):
                    taped_run_pmmh(context, st, alg._proposal, prop_dist, proposal_filter, sub_context, y, mutate_kernel=True)
                rec.emit("final", theta=context.stack_parameters(constrained=True), ll=st.filter_state.loglikelihood,
                         filter_means=st.filter_state.filter_means, kernel_loc=prop_dist.mean, kernel_scale=prop_dist.stddev)

        kinds = [k for k, _ in rec.events]
        path = os.path.join(GOLDEN, f"{name}.npz")
        np.savez_compressed(path, **flatten(rec.events))
        summary = {k: kinds.count(k) for k in dict.fromkeys(kinds)}
        extra = ""
        if "pmmh" not in case:
            outcomes = [str(f["outcome"]) for k, f in rec.events if k == "rejuvenated"]
            extra = f" outcomes={outcomes}"
        acc = [float(f["accepted"].double().mean()) for k, f in rec.events if k == "pmmh_accept"]
        print(f"wrote {path}: {summary}{extra} acceptance={[round(a, 2) for a in acc]}")


if __name__ == "__main__":
    main()
