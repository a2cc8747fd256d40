"""One particle marginal Metropolis-Hastings move for B parallel chains / theta-particles
(``pyfilter/inference/batch/mcmc/utils.py:14-77``, ``proposals/symmetric_mh.py``): propose theta*, re-filter the whole
data set with it (ONE fused ``batch_filter`` call for all B filters - a replayed hipGraph), accept per filter, swap the
accepted filters in with the column-move kernels.  Everything stays on the device: there is no host branch per move."""
import torch
from torch.distributions import Distribution, Independent, Normal

from .utils import construct_mvn, theta_normalize


class GaussianKernel:
    """``N(loc, scale_tril scale_tril^T)`` shared by all theta-particles, as ``ops.theta_fit`` produced it: the two tensors,
    read by the theta kernels (``ops.theta_propose`` / ``theta_accept``).  The ``torch.distributions`` surface the callers of
    ``SymmetricMH.build`` use (``loc / mean / scale_tril / batch_shape / event_shape / log_prob / sample``) goes through a
    ``MultivariateNormal`` built on demand."""

    def __init__(self, loc: torch.Tensor, scale_tril: torch.Tensor):
        self.loc, self.scale_tril = loc, scale_tril
        self.batch_shape, self.event_shape = torch.Size([]), torch.Size([loc.shape[0]])

    mean = property(lambda self: self.loc)

    def as_torch(self) -> Distribution:
        from torch.distributions import MultivariateNormal

        return MultivariateNormal(self.loc, scale_tril=self.scale_tril, validate_args=False)

    def log_prob(self, x):
        return self.as_torch().log_prob(x)

    def sample(self, size=torch.Size([])):
        return self.as_torch().sample(size)

    def __getattr__(self, name):
        # anything else of the ``torch.distributions`` surface (``covariance_matrix``, ``rsample``, ``entropy``, ...): a user
        # kernel or a trace consumer written against the reference's ``MultivariateNormal`` finds it here
        if name.startswith("__"):
            raise AttributeError(name)
        return getattr(self.as_torch(), name)


class SymmetricMH:
    """The proposal of the SMC^2 paper (``proposals/symmetric_mh.py``): a Gaussian fitted to the weighted theta-particles
    (unconstrained space), Cholesky factor scaled by 1.1.  Sharded runs fit it to ALL theta-particles (all-gather of
    ``(B, P)`` values and weights), so every rank holds the same kernel.  Scalar priors of the standard families on one
    GPU (``ThetaParticles.native_priors``): the fit is ONE launch (``ops.theta_fit``) and the move's theta arithmetic two
    more (``run_pmmh``)."""

    SCALE = 1.1

    def build(self, theta, state, filter_, y):  # -> Distribution | GaussianKernel (same read surface)
        values = theta.stack_parameters(constrained=False)
        weights_log = state.w
        shard = getattr(theta, "shard", None)
        if shard is not None and shard.collective:
            values, weights_log = shard.all_gather(values), shard.all_gather(weights_log)
        if theta.native_priors() is not None and weights_log.dtype == values.dtype:
            from .. import ops

            return GaussianKernel(*ops.theta_fit(values, weights_log, self.SCALE))
        return construct_mvn(values, theta_normalize(weights_log), scale=self.SCALE)

    def exchange(self, latest, candidate, mask) -> None:
        return


class RandomWalk:
    """The default proposal of PMMH (``proposals/random_walk.py``): theta* ~ N(theta, scale) around the chain's current
    (unconstrained) value; after an accepted move the kernel is re-centred in place (``exchange``)."""

    def __init__(self, scale=1e-2):
        self._scale = scale

    def build(self, theta, state, filter_, y):  # -> Distribution | GaussianKernel (same read surface)
        loc = theta.stack_parameters(constrained=False).clone()  # (re-centred in place after accepted moves: a tensor of its own)
        scale = torch.as_tensor(self._scale, device=loc.device, dtype=loc.dtype).expand_as(loc).clone()
        return Independent(Normal(loc, scale, validate_args=False), 1, validate_args=False)

    def exchange(self, latest, candidate, mask) -> None:
        m = mask.unsqueeze(-1)
        latest.base_dist.loc.copy_(torch.where(m, candidate.base_dist.loc, latest.base_dist.loc))
        latest.base_dist.scale.copy_(torch.where(m, candidate.base_dist.scale, latest.base_dist.scale))


class ThetaDraws:
    """Source of the theta-level random numbers of SMC^2 / PMMH (the reference takes them from torch's global generator:
    ``kernels/mh.py:53`` the resampling uniform, ``mcmc/utils.py:48`` the proposal's standard normals, ``:69`` the
    acceptance uniforms).  One CPU generator stream: every rank draws the numbers of ALL theta-particles in the same order
    and keeps its block, so a run's numbers do not depend on how many GPUs share it.  The parity tests substitute a
    replaying subclass that returns the reference's own (recorded) draws."""

    def __init__(self, generator: torch.Generator):
        self.generator = generator

    def uniform(self, shape) -> torch.Tensor:
        return torch.rand(tuple(shape), generator=self.generator, dtype=torch.float64)

    def normal(self, shape) -> torch.Tensor:
        return torch.randn(tuple(shape), generator=self.generator, dtype=torch.float64)


def as_draws(source):
    """``None`` (torch's global generator, like the reference), a ``torch.Generator`` or a ``ThetaDraws``."""
    if source is None or isinstance(source, ThetaDraws):
        return source
    return ThetaDraws(source)


def _to_device(host: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    """Host draws -> the device of ``like`` without making the host wait: staged in pinned memory (torch's caching host
    allocator), copied asynchronously on the current stream (a pageable source would be a synchronous copy)."""
    host = host.to(like.dtype)
    if like.is_cuda and not host.is_cuda:
        staged = torch.empty(host.shape, dtype=host.dtype, pin_memory=True)
        staged.copy_(host)
        return staged.to(like.device, non_blocking=True)
    return host.to(like.device)


def _draw(kernel: Distribution, size, shard, draws):
    """theta* ~ kernel (``mcmc/utils.py:48``: ``proposal_kernel.sample(size)``): ``(B, P)`` for a kernel shared by all
    filters (``size = (B,)``: SMC^2's Gaussian fit) and for a per-filter kernel (``size = ()``, ``batch_shape = (B,)``:
    the random walk) alike."""
    if draws is None:
        return kernel.sample(size)
    batched = len(kernel.batch_shape) > 0
    local = kernel.batch_shape[0] if batched else (size[0] if len(size) else 1)
    total = shard.total if shard is not None else local
    eps = draws.normal((total,) + tuple(kernel.event_shape))
    if shard is not None:
        eps = shard.slice(eps)
    loc = kernel.mean if not hasattr(kernel, "loc") else kernel.loc
    eps = _to_device(eps, loc)
    if hasattr(kernel, "scale_tril"):
        rvs = loc + (kernel.scale_tril @ eps.unsqueeze(-1)).squeeze(-1)
    else:  # a diagonal kernel: Independent(Normal(loc, scale), 1) - the random walk
        rvs = loc + kernel.stddev * eps
    return rvs if (len(size) or batched) else rvs[0]


def _uniforms(like: torch.Tensor, shard, draws):
    if draws is None:
        return torch.rand(like.shape, device=like.device, dtype=like.dtype)
    total = shard.total if shard is not None else like.shape[0]
    u = draws.uniform((total,))
    if shard is not None:
        u = shard.slice(u)
    return _to_device(u, like)


def _refilter(proposal_filter, y, stats):
    """The proposals' filters over the data.  A run on the column-cluster kernel reports through a status word when a launch could
    not make progress (its log-likelihoods are NaN then: the acceptance step rejects those proposals); verifying it here would
    make every move wait for its own re-filter, so the word goes to the caller (``stats["cluster_watch"]``), who reads it where
    it next waits for the device - ``watch_refilters`` - and without a ``stats`` dictionary the run is verified on the spot."""
    if stats is None or not hasattr(proposal_filter, "_cluster_gave_up"):
        return proposal_filter.batch_filter(y, bar=False)
    proposal_filter._defer_status_once = True
    try:
        res = proposal_filter.batch_filter(y, bar=False)
    finally:
        proposal_filter._defer_status_once = False  # (a subclass that does not pass through ParticleFilter.batch_filter; an error)
    watch = getattr(res, "_cluster_watch", None)
    if watch is not None:
        stats.setdefault("cluster_watch", []).append((watch[0], watch[1], proposal_filter))
    return res


def watch_refilters(stats) -> int:
    """Call where the host has just waited for the device: how many of the re-filters noted in ``stats`` had a column-cluster launch
    give up (their proposals were rejected: NaN log-likelihoods) - the status words are cleared and the filters told, so that
    the caller can repeat the move (SMC^2) or count it (PMMH)."""
    gave_up = 0
    for status, plan, filt in (stats or {}).pop("cluster_watch", []):
        if int(status.item()) != 0:
            filt._cluster_gave_up(plan)
            gave_up += 1
    return gave_up


def _run_pmmh_native(theta, state, proposal, kernel: "GaussianKernel", proposal_filter, proposal_theta, y, draws, trace, stats,
                     overlap=None):
    """``run_pmmh`` for a Gaussian kernel shared by all theta-particles and scalar priors of the standard families: the
    same move with its theta arithmetic in three launches (``csrc/pf_theta.hpp``) - theta* with its constrained values and
    log prior, the reverse kernel's fit, the acceptance step - instead of ~130 small torch launches the re-filter waits
    behind."""
    from .. import ops

    mark = (stats or {}).get("mark") or (lambda label: None)
    priors = theta.native_priors()
    b, p = theta.batch_shape[0], priors.P
    like = kernel.loc
    shard = getattr(theta, "shard", None)
    if draws is not None:  # (every rank draws the numbers of ALL theta-particles and keeps its block: see ThetaDraws)
        eps = draws.normal((shard.total if shard is not None else b, p))
        eps = _to_device(shard.slice(eps) if shard is not None else eps, like)
    else:
        eps = torch.randn((b, p), device=like.device, dtype=like.dtype)
    rvs, prior_star = ops.theta_propose(priors, kernel.loc, kernel.scale_tril, eps, [proposal_theta[n] for n in proposal_theta.names()],
                                        prior_out=proposal_theta._prior_row())
    proposal_theta.adopt_proposal(rvs, prior_star)
    mark("  theta* proposed")
    proposal_filter.initialize_model(proposal_theta)  # (rebuilt from theta*: see run_pmmh)
    mark("  model rebuilt")
    new_res = _refilter(proposal_filter, y, stats)
    mark("  re-filter issued")
    if overlap is not None:
        overlap()
    new_kernel = proposal.build(proposal_theta, state.replicate(new_res), proposal_filter, y)
    log_acc, accepted, rate = ops.theta_accept(
        theta.stack_parameters(constrained=False), rvs, (kernel.loc, kernel.scale_tril), (new_kernel.loc, new_kernel.scale_tril),
        theta.eval_priors(constrained=False), prior_star, state.filter_state.loglikelihood, new_res.loglikelihood,
        _uniforms(prior_star, shard, draws))
    mark("  acceptance issued")
    if stats is not None:
        stats["rate"] = rate
    if trace is not None:
        trace.append(dict(kind="pmmh", rvs=rvs, proposed_ll=new_res.loglikelihood.clone(), log_acc=log_acc, accepted=accepted,
                          new_kernel=new_kernel))
    state.filter_state.exchange(new_res, accepted)
    theta.exchange(proposal_theta, accepted)
    mark("  accepted filters swapped in")
    return accepted


def run_pmmh(theta, state, proposal, proposal_kernel: Distribution, proposal_filter, proposal_theta, y: torch.Tensor,
             size=torch.Size([]), mutate_kernel: bool = False, generator=None, trace=None, stats=None, overlap=None) -> torch.Tensor:
    """One PMMH iteration (``mcmc/utils.py:14-77``).  ``theta`` / ``state``: the chains' parameters and algorithm state
    (``state.filter_state`` a ``FilterResult``); ``proposal_filter`` reads ``proposal_theta``.  Returns the ``(B,)``
    boolean mask of accepted proposals; ``state`` and ``theta`` are updated in place.  ``trace``: an optional list that
    receives the move's intermediate quantities (references, no copies, no synchronisation) - diagnostics, and what the
    parity tests compare with the reference's own values.  ``stats``: an optional dict; the native route leaves the
    move's acceptance rate (a device scalar) under ``"rate"``.  ``overlap``: called once the re-filter has been issued -
    host work of the caller that the move's outcome does not depend on until the filters are compared (SMC^2 moves the
    resampled filters' states there: the device is busy with the re-filter meanwhile)."""
    shard = getattr(theta, "shard", None)
    draws = as_draws(generator)
    if (isinstance(proposal_kernel, GaussianKernel) and isinstance(proposal, SymmetricMH) and not mutate_kernel and
            theta.native_priors() is not None and proposal_theta.native_priors() is not None):
        return _run_pmmh_native(theta, state, proposal, proposal_kernel, proposal_filter, proposal_theta, y, draws, trace, stats,
                                overlap)
    rvs = _draw(proposal_kernel, size, shard, draws)
    proposal_theta.unstack_parameters(rvs, constrained=False)
    # the model is REBUILT from theta* (mcmc/utils.py:52-53): whatever the builder derives from the parameters - e.g. the
    # stationary initial distribution of an Ornstein-Uhlenbeck process - belongs to the proposed values, not the old ones
    proposal_filter.initialize_model(proposal_theta)
    new_res = _refilter(proposal_filter, y, stats)
    if overlap is not None:
        overlap()

    diff_logl = new_res.loglikelihood - state.filter_state.loglikelihood
    diff_prior = proposal_theta.eval_priors(constrained=False) - theta.eval_priors(constrained=False)
    new_kernel = proposal.build(proposal_theta, state.replicate(new_res), proposal_filter, y)
    current = theta.stack_parameters(constrained=False)
    diff_prop = new_kernel.log_prob(current) - proposal_kernel.log_prob(rvs)

    log_acc = diff_prop + diff_prior + diff_logl
    accepted = _uniforms(log_acc, shard, draws).log() < log_acc  # (NaN compares False: a failed proposal is rejected)

    if trace is not None:
        trace.append(dict(kind="pmmh", rvs=rvs, proposed_ll=new_res.loglikelihood.clone(), log_acc=log_acc, accepted=accepted,
                          new_kernel=new_kernel))
    state.filter_state.exchange(new_res, accepted)
    theta.exchange(proposal_theta, accepted)
    if mutate_kernel:
        proposal.exchange(proposal_kernel, new_kernel, accepted)
    return accepted


class PMMHState:
    """Algorithm state of ``PMMH`` (``mcmc/state.py``): the chains' current filter results and every sample drawn so far -
    one preallocated ``(num_samples + 1, B, P)`` device buffer written in place (constrained values, the parameters side
    by side as in ``ThetaParticles.stack_parameters``), not a tensor re-concatenated per move."""

    def __init__(self, filter_state, first: torch.Tensor, num_samples: int):
        self.filter_state = filter_state
        self.w = torch.zeros(first.shape[0], device=first.device, dtype=first.dtype)  # (equal weights: the chains are iid)
        self.chain = first.new_empty((num_samples + 1,) + tuple(first.shape))
        self.chain[0] = first
        self.length = 1
        self.accepted = torch.zeros(first.shape[0], device=first.device, dtype=first.dtype)

    def replicate(self, filter_state) -> "PMMHState":
        other = PMMHState.__new__(PMMHState)
        other.filter_state, other.w, other.chain, other.length, other.accepted = filter_state, self.w, self.chain, self.length, self.accepted
        return other

    def update_chain(self, sample: torch.Tensor, accepted: torch.Tensor):
        self.chain[self.length] = sample
        self.length += 1
        self.accepted += accepted.to(self.accepted.dtype)

    @property
    def samples(self) -> torch.Tensor:
        """``(draws so far, B, P)``: the chains, initial values included."""
        return self.chain[: self.length]

    def acceptance_rate(self) -> torch.Tensor:
        return self.accepted / max(1, self.length - 1)


class PMMH:
    """Particle marginal Metropolis-Hastings with ``num_chains`` parallel chains on the filter's batch dimension
    (``inference/batch/mcmc/pmmh.py``): every move re-filters the whole data set for all chains in one fused
    ``batch_filter`` call (a replayed hipGraph from the second move on) and nothing in the loop waits for the device.
    ``filter_`` is built with a model builder ``theta -> StateSpaceModel``; ``priors`` maps names to distributions.
    ``initializer="mean"``: the chains start at the priors' means (Monte-Carlo means where a prior has no closed form)."""

    MONTE_CARLO_SAMPLES = 10_000

    def __init__(self, filter_, num_samples: int, priors, num_chains: int = 4, proposal=None, initializer: str = "mean",
                 device="cuda", dtype=torch.float32, seed: int = 0):
        from .parameters import ThetaParticles

        if initializer != "mean":
            raise NotImplementedError(f"``{initializer}`` is not configured!")
        self.filter = filter_
        self.num_samples = int(num_samples)
        self.theta = ThetaParticles(priors, num_chains, device, dtype)
        self.filter.set_batch_shape(torch.Size([num_chains]))
        self._proposal = proposal or RandomWalk()
        self._gen = torch.Generator().manual_seed(seed)

    def initialize(self, y: torch.Tensor) -> PMMHState:
        self.theta.initialize_parameters(self._gen)
        for name, prior in self.theta.priors.items():
            d = prior.distribution
            try:
                mean = d.mean
                if not torch.isfinite(mean).all():
                    raise NotImplementedError
            except NotImplementedError:
                with torch.random.fork_rng(devices=[self.theta.device] if self.theta.device.type == "cuda" else []):
                    torch.manual_seed(int(torch.randint(0, 2 ** 62, (), generator=self._gen)))
                    mean = d.sample(torch.Size([self.MONTE_CARLO_SAMPLES])).mean(dim=0)
            self.theta[name].copy_(mean.to(self.theta[name]).expand_as(self.theta[name]))
        self.filter.initialize_model(self.theta)
        first = self.filter.batch_filter(y, bar=False)
        return PMMHState(first, self.theta.stack_parameters(True), self.num_samples)

    def fit(self, y: torch.Tensor) -> PMMHState:
        state = self.initialize(y)
        kernel = self._proposal.build(self.theta, state, self.filter, y)
        proposal_theta = self.theta.like()
        proposal_filter = self.filter.copy()
        proposal_filter.initialize_model(proposal_theta)
        stats = {}
        for _ in range(self.num_samples):
            accepted = run_pmmh(self.theta, state, self._proposal, kernel, proposal_filter, proposal_theta, y,
                                self.filter.batch_shape, mutate_kernel=True, generator=self._gen, stats=stats)
            state.update_chain(self.theta.stack_parameters(True), accepted)
            if len(stats.get("cluster_watch", ())) >= 64:  # (the words are sticky per plan: looked at every 64 moves and at the end)
                self.unfinished_moves = getattr(self, "unfinished_moves", 0) + watch_refilters(stats)
        # moves whose re-filter was cut short by a column-cluster launch that gave up count as rejections (pf_amd.h: PF_ROUTE_CLUSTER)
        self.unfinished_moves = getattr(self, "unfinished_moves", 0) + watch_refilters(stats)
        return state
