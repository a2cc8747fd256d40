"""SMC^2 (Chopin, Jacob & Papaspiliopoulos) on the hot path: theta-particles on the filters' batch dimension, one fused
``filter()`` move per observation, an ESS-triggered rejuvenation that resamples whole filters and moves them with PMMH.

Mirrors ``pyfilter/inference/sequential/smc2.py:53-65`` (``SMC2._step``), ``sequential/state.py:35-44`` (weights / ESS
bookkeeping), ``sequential/kernels/mh.py:52-140`` (``ParticleMetropolisHastings.update`` incl. the particle doubling of
``_increase_states``) - with the reference's ``InferenceContext`` replaced by ``ThetaParticles``.

More than one GPU (``Shard``; one process per GPU): every rank owns a block of the theta-particles and runs the same code
on it.  Per observation the per-theta log-likelihood increments are all-gathered (B floats) so that all ranks hold the
same theta-weights and take the same ESS decision; a rejuvenation performs the same systematic resampling of the
theta-particles on every rank (same weights, same uniform), takes the surviving filters' states from their owners
(``FilterResult`` blocks via ``Shard.take``), fits the proposal to all theta-particles and averages the acceptance rate
over all of them."""
import time
from typing import Callable, Optional

import torch

from ..distributed import Shard
from ..filters.result import FilterResult
from .parameters import ThetaParticles
from .pmmh import SymmetricMH, as_draws, run_pmmh, watch_refilters
from .. import ops as _ops
from ..hints import HINTS
from ..hints import HINTS as _HINTS
from .utils import theta_ess, theta_normalize, theta_systematic


class TooManyIncreases(Exception):
    pass


def _theta_stats(gw: torch.Tensor) -> torch.Tensor:
    """``(..., B)`` theta log-weights -> ``(..., 2)``: ESS and an "every weight finite" flag per row - on the GPU one launch
    (``pf_theta_ess``), plain torch for CPU tensors (the multi-process CPU tests)."""
    if gw.is_cuda:
        from .. import ops

        return ops.theta_ess(gw)
    rows = gw.reshape(-1, gw.shape[-1])
    out = torch.stack([torch.stack([theta_ess(r), r.isfinite().all().to(r.dtype)]) for r in rows])
    return out.reshape(gw.shape[:-1] + (2,))


def _theta_path_native(w: torch.Tensor, ll: torch.Tensor, shard) -> bool:
    from ..hints import HINTS

    return bool(ll.is_cuda and (shard is None or not shard.collective) and HINTS.theta_kernels and ll.dim() == 2 and ll.dtype == w.dtype)


def _theta_path(w: torch.Tensor, ll: torch.Tensor, shard, rows=None, status=None):
    """The theta log-weights after each of a block's ``n`` observations, ``w + ll.cumsum(0)``, and their ``(n, 2)`` statistics
    (``_theta_stats``) - one launch on one GPU (``pf_theta_path``; ``rows`` / ``status``: the statistics also land in host memory,
    ``ops.HostRows``); sharded, the rows are all-gathered in between."""
    if _theta_path_native(w, ll, shard):
        from .. import ops

        return ops.theta_path(w.contiguous(), ll.contiguous(), rows, status)
    assert rows is None
    w_path = w + ll.cumsum(0)
    return w_path, _theta_stats(shard.all_gather(w_path, dim=1) if shard is not None and shard.collective else w_path)


class SMC2State:
    """Algorithm state (``sequential/state.py:8-95``): theta log-weights ``w`` (this rank's block), the filters' result,
    the ESS history and the observations parsed so far (PMMH re-filters them)."""

    def __init__(self, weights: torch.Tensor, filter_state: FilterResult, shard: Optional[Shard] = None):
        self.w = weights
        self._online = None  # the fast driver of a step() loop (ParticleFilter.online_run): rows / latest state pending in its arrays
        self.filter_state = filter_state
        self.shard = shard
        self.ess = [self._ess()]
        self.parsed = []
        self.current_iteration = 0
        self._series = None  # ``fit``: (the series being parsed, host flags "observation is not all-NaN") - see parsed_data

    @property
    def filter_state(self) -> FilterResult:
        """The filters' result - brought up to date first when an observation-by-observation loop keeps its latest moves in the
        fast driver's arrays (``_OnlineRun.flush``)."""
        run = self.__dict__.get("_online")
        if run is not None:
            run.flush()
        return self._filter_state

    @filter_state.setter
    def filter_state(self, value):
        self._filter_state = value
        self._online = None  # (another result: the driver of the previous one has nothing to say about it)

    def __getstate__(self):
        """Copies / pickles carry the result, not the driver's launch arguments (raw device addresses of this process)."""
        state = dict(self.__dict__)
        run = state.pop("_online", None)
        if run is not None:
            run.flush()
        state["_online"] = None
        state.pop("_online_na", None)
        return state

    def global_weights(self) -> torch.Tensor:
        return self.w if self.shard is None or not self.shard.collective else self.shard.all_gather(self.w)

    def _ess(self) -> torch.Tensor:
        """ESS of ALL theta-weights; ``self.stats`` keeps it together with the "every weight finite" flag - on the GPU both
        come out of one launch (``pf_theta_ess``), so an observation costs one small device -> host copy."""
        self.stats = _theta_stats(self.global_weights())
        return self.stats[0]

    def _theta_step_applies(self) -> bool:
        """``append`` updates the weights with ``pf_theta_step`` (one launch; the host slot, the running total and a move's status
        word can ride along): one GPU's own block of contiguous weights, no collective, theta kernels enabled."""
        w = self.w
        return bool(w.is_cuda and (self.shard is None or not self.shard.collective) and w.dim() == 1 and w.is_contiguous()
                    and _HINTS.theta_kernels)

    def append(self, filter_state, slot=None, acc: Optional[torch.Tensor] = None, status: Optional[torch.Tensor] = None) -> bool:
        """``w += ll_t`` and the new ESS (state.py:35-44).  Sharded: the all-gather of the increments happens here.
        ``slot`` (an ``ops.HostSlot``): the statistics also land in host memory - returns True when they will.  ``acc`` /
        ``status``: the filters' running log-likelihood to add ``ll_t`` to as well and the status word of the move that produced
        it (``ops.theta_step``: a non-zero word leaves everything as it was and shows in ``slot.status``)."""
        ll = filter_state.get_loglikelihood()
        w = self.w
        if self._theta_step_applies() and ll.dtype == w.dtype and ll.shape == w.shape and ll.is_contiguous():
            # the update and its statistics in ONE launch (pf_theta_step)
            self.stats = _ops.theta_step(w, ll, slot, acc=acc, status=status)
            self.ess.append(self.stats[0])
            return slot is not None
        assert status is None, "a watched move needs the pf_theta_step route (SMC2._step checks _theta_step_applies first)"
        if acc is not None:
            acc += ll
        self.w += ll
        self.ess.append(self._ess())
        return False

    def append_data(self, y: torch.Tensor):
        self.parsed.append(y)

    def attach_series(self, y: torch.Tensor, flags: Optional[torch.Tensor]):
        """``fit`` knows the whole series up front: the observations parsed so far are a PREFIX of it - a view instead of a
        stack of per-observation tensors rebuilt at every rejuvenation - and so are their host flags (which PMMH's
        re-filter would otherwise fetch from the device once per rejuvenation)."""
        self._series = (y, flags)

    @property
    def parsed_data(self) -> torch.Tensor:
        series = getattr(self, "_series", None)
        if series is not None:
            return series[0][: len(self.parsed)]
        return torch.stack(self.parsed, dim=0)

    def parsed_flags(self) -> Optional[torch.Tensor]:
        series = getattr(self, "_series", None)
        return None if series is None or series[1] is None else series[1][: len(self.parsed)]

    def normalized_weights(self) -> torch.Tensor:
        return theta_normalize(self.global_weights())

    def replicate(self, filter_state) -> "SMC2State":
        other = SMC2State.__new__(SMC2State)
        other.w, other.filter_state, other.shard = torch.zeros_like(self.w), filter_state, self.shard  # (the setter clears _online)
        other.ess, other.parsed, other.current_iteration, other.stats = [], self.parsed, self.current_iteration, None
        other._series = getattr(self, "_series", None)
        return other

    def state_dict(self):
        """The reference's wire format (``inference/state.py:39-48``, ``sequential/state.py:56-66``, ``container.py:113-126``):
        the ESS history and the parsed observations as unbounded tensor deques, the filters' result, the theta log-weights
        (this rank's block) and the iteration counter."""
        from collections import OrderedDict

        series = OrderedDict([("tensor_deque_None__ess", torch.stack(self.ess)),
                              ("tensor_deque_None__parsed_data", self.parsed_data if self.parsed else torch.tensor([]))])
        return OrderedDict([("tensor_tuples", series), ("filter_state", self.filter_state.state_dict()), ("w", self.w),
                            ("current_iteration", self.current_iteration)])

    def load_state_dict(self, state_dict):
        """Continues from a serialised run (``tests/inference/test_sequential.py:55-93``): the receiving state is a freshly
        initialised one of the same shape."""
        series = state_dict["tensor_tuples"]
        self.ess = list(series["tensor_deque_None__ess"].to(self.w.device).unbind(0))
        parsed = series["tensor_deque_None__parsed_data"]
        self.parsed = list(parsed.to(self.w.device).unbind(0)) if parsed.numel() else []
        self.filter_state.load_state_dict(state_dict["filter_state"])
        self.w = state_dict["w"].to(self.w.device)
        self._series = None
        self.current_iteration = int(state_dict["current_iteration"])
        self.stats = None


def _take_filters(result: FilterResult, shard: Optional[Shard], mine: torch.Tensor, route=None):
    """``FilterResult.resample`` for a sharded set of filters: ``mine`` = the GLOBAL ancestors of this rank's positions
    (single process: all of them, and the reference's in-place gather - one ``pf_columns_gather`` per buffer)."""
    if shard is None or not shard.collective:
        result.resample(mine)
        return
    route = route or shard.route(mine)  # who sends which columns to whom: built once, every buffer moves through it
    # every per-filter quantity of the result travels as its block along the batch dimension
    result._loglikelihood.copy_(route.take(result._loglikelihood))
    log = result._moments
    if log._buf is not None:
        cols = log._live_columns()  # (1, B_local, rows * 2 dim)
        log._buf = route.take(cols[0]).reshape(log._buf.shape)
    from .. import ops

    for s in result._states:
        s._ensure_moments()
        ts = s.timeseries_state
        x = ops.to_soa(ts.value, True, ts.value.dim() > 2)              # (D, B_local, N)
        s["_x"] = ts.copy(values=ops.from_soa(route.take(x, dim=1).contiguous(), True, ts.value.dim() > 2))
        s["_w"] = ops.from_cols(route.take(ops.to_cols(s["_w"])).contiguous(), True)
        s["_prev_inds"] = ops.from_cols(route.take(ops.to_cols(s["_prev_inds"])).contiguous(), True)
        for k in ("_ll", "_mean", "_var"):
            s[k] = route.take(s[k])
    return route


class ParticleMetropolisHastings:
    """The rejuvenation kernel (``kernels/mh.py:15-140``)."""

    OVERLAP_FILTER_MOVE = True  # (development switch: False moves the resampled filters before the first move, like the reference)

    def __init__(self, num_steps: int = 1, proposal=None, distance_threshold: Optional[float] = None,
                 acceptance_threshold: float = 0.2, max_increases: int = 5, resampler: Callable = theta_systematic):
        self._n_steps = num_steps
        self._proposal = proposal or SymmetricMH()
        self._dist_thresh = distance_threshold
        self._is_adaptive = distance_threshold is not None
        self._acceptance_threshold = acceptance_threshold
        self._max_increases = max_increases
        self._increases = 0
        self._resampler = resampler
        self.acceptance_history = []
        self.trace = None  # set to a list to collect every update's intermediate quantities (see ``run_pmmh``)
        self.timeline = None  # set to a list: host wall-clock marks (label, seconds) of every update (tools/smc2_small.py)

    def update(self, theta: ThetaParticles, filter_, state: SMC2State, generator=None) -> SMC2State:
        shard = state.shard
        sharded = shard is not None and shard.collective
        mark = (lambda label: self.timeline.append((label, time.perf_counter()))) if self.timeline is not None else (lambda label: None)
        mark("start")
        # the same resampling on every rank: same (gathered) weights, same uniform (the generator is a CPU stream seeded
        # identically everywhere and advanced in lock step - no broadcast needed)
        draws = as_draws(generator)
        u = draws.uniform(()) if draws is not None else torch.rand(())
        gw = state.global_weights()
        if self._resampler is theta_systematic and gw.is_cuda and HINTS.theta_kernels:
            from .. import ops

            indices = ops.theta_resample(gw, float(u))  # (normalisation, cdf and search in one launch)
        else:
            indices = self._resampler(theta_normalize(gw), u)
        mine = shard.slice(indices) if sharded else indices
        data, data_flags = state.parsed_data, state.parsed_flags()
        dist = self._proposal.build(theta, state, filter_, data)

        if self.trace is not None:
            self.trace.append(dict(kind="rejuvenate", indices=indices, kernel=dist))
        mark("resampled + proposal fitted")
        # one exchange plan for the parameters and the filters' states (every rank holds ALL ancestors: no all-gather)
        route = shard.route(mine, full_index=indices) if sharded else None
        theta.resample(mine, route)
        # the surviving filters' states are not read before the first move compares log-likelihoods: they move while the
        # device runs that move's re-filter (``run_pmmh(overlap=...)``) - 0.4 ms of host work off the serial path
        unmoved = [True]

        def move_filters():
            if unmoved[0]:
                unmoved[0] = False
                _take_filters(state.filter_state, shard, mine, route)
                mark("filters moved")

        if not self.OVERLAP_FILTER_MOVE:
            move_filters()
        shape = torch.Size([]) if any(dist.batch_shape) else filter_.batch_shape

        old = theta.stack_parameters(constrained=False)
        proposal_theta = theta.like()
        proposal_filter = filter_.copy()  # (``run_pmmh`` builds its model from theta* before it filters)
        if data_flags is not None and hasattr(proposal_filter, "_obs_cache"):
            proposal_filter._obs_cache = (data, data._version, data_flags)  # (known on the host: no device round trip)

        previous_distance, acceptance_rate = 0.0, 0.0
        for i in range(self._n_steps):
            stats = {"mark": mark} if self.timeline is not None else {}
            # (what a repeated move starts from again: the theta-level stream and the proposals' filter draw epoch)
            gen = getattr(draws, "generator", None)
            rewind = (gen.get_state() if isinstance(gen, torch.Generator) else None, getattr(proposal_filter, "_draws", None))
            accepted = run_pmmh(theta, state, self._proposal, dist, proposal_filter, proposal_theta, data,
                                shape, mutate_kernel=False, generator=draws, trace=self.trace, stats=stats, overlap=move_filters)
            if "rate" in stats:  # (the native theta route: the acceptance kernel counted this rank's share)
                rate = stats["rate"]
                if sharded:
                    rate = shard.all_mean(rate * accepted.numel(), accepted.numel())
            else:
                rate = accepted.float().sum()
                rate = shard.all_mean(rate, accepted.numel()) if sharded else rate / accepted.numel()
            mark("move issued")
            rate_now = float(rate)  # the kernel's one host decision per move
            if watch_refilters(stats):
                # the move's re-filter ran on the column-cluster kernel and a launch gave up (its proposals were rejected, nothing
                # else happened): the move again - the filter now re-issues on the per-step route itself (its plan was told)
                proposal_filter._per_step_once = True
                if rewind[0] is not None:
                    gen.set_state(rewind[0])  # the same proposals, the same acceptance uniforms ...
                if rewind[1] is not None:
                    proposal_filter._draws = rewind[1]  # ... the same particle draws: the move the launch cut short, not another one
                stats.pop("rate", None)
                accepted = run_pmmh(theta, state, self._proposal, dist, proposal_filter, proposal_theta, data,
                                    shape, mutate_kernel=False, generator=draws, trace=self.trace, stats=stats, overlap=move_filters)
                rate = stats["rate"] if "rate" in stats else accepted.float().mean()
                if sharded:
                    rate = shard.all_mean(rate * accepted.numel(), accepted.numel())
                rate_now = float(rate)
            acceptance_rate = (rate_now + i * acceptance_rate) / (i + 1)
            mark("move done on the device")
            self.acceptance_history.append(acceptance_rate)
            if acceptance_rate < self._acceptance_threshold:  # abort early: more state particles are needed
                return self._increase_states(filter_, state, theta)
            if not self._is_adaptive:
                continue
            new = theta.stack_parameters(constrained=False)
            reach = (new - old).abs().amax(dim=0)  # (P,): the largest move per parameter over the theta-particles ...
            if sharded:
                reach = shard.all_max(reach)       # ... over ALL of them (mh.py:95: norm(dim=0, p=inf) of the full set)
            distance = float(reach.mean())
            if abs(distance - previous_distance) <= self._dist_thresh * previous_distance:
                break
            previous_distance = distance

        move_filters()  # (num_steps = 0: a pure resampling)
        filter_.initialize_model(theta)
        state.w.fill_(0.0)
        mark("end")
        return state

    def _increase_states(self, filter_, state: SMC2State, theta: ThetaParticles) -> SMC2State:
        """Doubles the number of state particles and re-filters the parsed data (``kernels/mh.py:110-140``)."""
        self._increases += 1
        if self._increases > self._max_increases:
            raise TooManyIncreases(f"Configuration only allows {self._max_increases}!")
        filter_.initialize_model(theta)
        filter_.increase_particles(2.0)
        filter_.set_batch_shape(filter_.batch_shape)
        data, data_flags = state.parsed_data, state.parsed_flags()
        if data_flags is not None and hasattr(filter_, "_obs_cache"):
            filter_._obs_cache = (data, data._version, data_flags)
        new_filter_state = filter_.batch_filter(data, bar=False)
        weight = new_filter_state.loglikelihood - state.filter_state.loglikelihood
        res = SMC2State(weight, new_filter_state, state.shard)
        res.ess, res.parsed, res.current_iteration = state.ess, state.parsed, state.current_iteration
        res._series = getattr(state, "_series", None)
        return res


class SMC2:
    """``SMC2(filter_, particles, threshold, kernel)`` (``smc2.py:11-65``).  ``filter_`` is built with a model *builder*
    ``theta -> StateSpaceModel`` (the reference's ``context -> model``); ``priors`` maps parameter names to distributions.
    ``particles`` is the TOTAL number of theta-particles; under a process group each rank holds its block."""

    def __init__(self, filter_, particles: int, priors, threshold: float = 0.2, kernel=None, max_increases: int = 5,
                 device="cuda", dtype=torch.float32, seed: int = 0, group=None, block: Optional[int] = None, **kwargs):
        self.filter = filter_
        self.shard = Shard(particles, group)
        self.particles = torch.Size([particles])
        self.theta = ThetaParticles(priors, self.shard.local, device, dtype, self.shard)
        self.filter.set_batch_shape(torch.Size([self.shard.local]))
        self._threshold = threshold
        # observations ``fit`` runs ahead of its rejuvenation test.  A block costs the host ~0.25 ms whatever its length, so it
        # should hold about that much device time: 32 moves of a few hundred thousand particles (1 000 theta x 400: 5 us per
        # move), 16 of more (128 x 8 192: 12 us per move; profiles/r04_smc2_block_sweep.txt)
        if block is None:
            work = self.shard.local * int(filter_.particles[0])
            # (2^23 particles and more - 1 024 theta x 8 192 on one GPU: a move is ~53 us of kernel, and what a rejuvenation inside a
            # block costs - the cut replay, the dropped speculative successor - outweighs the host time a longer block saves:
            # tools/smc2_timeline.py, 8: 53.5 ms, 16: 56.7, 32: 62.4)
            block = 32 if work < (3 << 18) else (16 if work < (1 << 23) else 8)
        self._block = max(1, int(block))
        self._kernel = ParticleMetropolisHastings(proposal=kernel, max_increases=max_increases, **kwargs)
        self._gen = torch.Generator().manual_seed(seed)  # CPU: the same stream on every rank (theta-level draws)
        self._seed = seed
        if self.shard.world > 1 and hasattr(self.filter, "_seed"):
            # the kernels key their Philox streams by the LOCAL column index: decorrelate the ranks' blocks
            self.filter._seed = (self.filter._seed + 0xD1B54A32D192ED03 * self.shard.rank) & 0xFFFFFFFFFFFFFFFF

    def initialize(self, theta0: Optional[torch.Tensor] = None) -> SMC2State:
        """Draws the theta-particles from their priors (``sequential/base.py:52-62``) - or starts from ``theta0``, the
        ``(B, P)`` stacked constrained values of ALL theta-particles (every rank keeps its block)."""
        g = torch.Generator().manual_seed(self._seed * 7919 + 13)  # every rank draws all B and keeps its block
        self.theta.initialize_parameters(g)
        if theta0 is not None:
            mine = self.shard.slice(theta0) if self.shard.collective else theta0
            self.theta.unstack_parameters(mine.to(device=self.theta.device, dtype=self.theta.dtype), constrained=True)
        self.filter.initialize_model(self.theta)
        init_state = self.filter.initialize()
        w = torch.zeros(self.shard.local, device=init_state.get_loglikelihood().device, dtype=init_state.get_loglikelihood().dtype)
        return SMC2State(w, self.filter.initialize_with_result(init_state), self.shard)

    def step(self, y: torch.Tensor, state: SMC2State) -> SMC2State:
        state = self._step(y, state)
        state.current_iteration += 1
        return state

    def _step(self, y: torch.Tensor, state: SMC2State) -> SMC2State:
        """One observation (``smc2.py:53-65``)."""
        state.append_data(y)
        # the reference's host branch (smc2.py:59-62) needs (ESS, all finite) on the host after every observation: the kernel
        # that updates the theta-weights writes the pair into host memory as well, and the host polls for it - no copy command
        slot = self.__dict__.get("_host_slot")
        if slot is None and state.w.is_cuda:
            try:
                slot = self._host_slot = _ops.HostSlot()
            except _ops.L.PfAmdError as e:  # (no coherent host memory to be had: the copy command per observation it is)
                import warnings

                warnings.warn(f"SMC2.step: no host slot ({e}); the statistics travel by a copy command per observation")
                slot = self._host_slot = False
        slot = slot or None
        # With the slot this loop reads something the device wrote after EVERY move - so the move may take the column-cluster
        # kernel, whose launches report instead of hanging when they cannot make progress (hints.py): the move's status word
        # rides through pf_theta_step into the slot, and a move that gave up is issued again on the per-step route.
        filt = self.filter
        # The fast driver (filters/particle/base.py: _OnlineRun): the loop's moves as pieces of ONE run on one argument block - per
        # observation one pf_filter_run call, one pf_theta_step call and the poll; the FilterResult catches up when somebody looks
        # (state.filter_state: the rejuvenation below, the caller).  Same kernels, same draws per seed as the path below.
        if slot is not None and state._theta_step_applies() and hasattr(filt, "online_run"):
            run = state._online
            if (run is None or run.filt is not filt or run.result is not state._filter_state) and \
                    state.__dict__.get("_online_na") is not state._filter_state:
                run = state._online = filt.online_run(state._filter_state)
                if run is None:
                    state._online_na = state._filter_state  # (asked once per result: the path below it is)
            if run is not None and isinstance(y, torch.Tensor) and y.numel() == run.o:  # (one observation row shared by the filters)
                ess, finite = run.observe(y, state.w, slot)
                stats, row = run.last_stats
                state.stats = stats[row]
                state.ess.append(state.stats[0])
                if ess < self._threshold * self.particles[0] or not finite:
                    state = self._kernel.update(self.theta, self.filter, state, generator=self._gen)
                return state
        watching = slot is not None and state._theta_step_applies() and hasattr(filt, "_online_cluster")
        if watching:
            filt._online_cluster = True
        try:
            filter_state = filt.filter(y, state.filter_state.latest_state, result=state.filter_state)
        finally:
            if watching:
                filt._online_cluster = False
        watched = getattr(filt, "_watched_move", None) if watching else None
        if watched is None:
            ess, finite = slot.wait() if state.append(filter_state, slot) else state.stats.tolist()
        else:
            total = state.filter_state._loglikelihood
            state.append(filter_state, slot, acc=total, status=watched[0])
            ess, finite = slot.wait()
            if slot.status:  # the launch gave up: nothing was added - the move again, on the per-step route, and its update
                watched[1]()
                state.ess.pop()
                state.append(filter_state, slot, acc=total)
                ess, finite = slot.wait()
        if ess < self._threshold * self.particles[0] or not finite:
            state = self._kernel.update(self.theta, self.filter, state, generator=self._gen)
        return state

    # ---- fit(): the filters run a block of observations ahead of the host's rejuvenation test ---------------------------
    # The reference tests the ESS of the theta-weights on the host after every observation (``smc2.py:59-62``) - a device
    # round trip per observation, which is what an SMC^2 run on a GPU spends its time on.  Here the filters run a whole
    # block as one fused sequence (``filter_block``); the block's theta-weight paths ``w + cumsum(ll)`` (one all-gather per
    # block when sharded) and their ESS come back in ONE small copy, and the test is applied to them in order.  No
    # rejuvenation due (the rule, not the exception): the block is committed.  Due at observation ``j``: the moves after it
    # never happened - the block is run again cut after ``j`` *on the same draws*, so state and weights belong to one
    # particle system - and the kernel takes over exactly where the reference's would.
    # Blocks are PIPELINED: block k + 1 is issued (from block k's final state, as if no rejuvenation were due) before the
    # host waits for block k's statistics, so the device never idles while the host prepares launches; when block k does
    # contain a rejuvenation, the speculative successor is simply dropped.
    def _issue_block(self, ys: torch.Tensor, flags: torch.Tensor, latest, w: torch.Tensor, slot: int, token=None, per_step: bool = False):
        """Issues one block from (``latest`` state, theta-weights ``w``); nothing here waits for the device.  Returns None
        when the filter has no fused block route.  ``token`` / ``per_step``: the block again on the draws of an earlier issue,
        on the per-step kernel route (what ``_verified`` asks for when a column-cluster launch gave up)."""
        filt, shard = self.filter, self.shard
        out = filt.filter_block(ys, latest, observed=flags, replay=token, per_step=per_step, defer_status=True)
        if out is None:
            return None
        res, ll, token = out
        status = getattr(res, "status", None)  # (the block took the column-cluster kernel: its status word travels with the statistics)
        # one GPU's own theta block: the statistics rows (and the status word) are written into host memory by the kernel that
        # computes them - the host polls them row by row; nothing is copied, no event is recorded behind the block
        rows = self._host_rows(slot, ll.shape[0]) if _theta_path_native(w, ll, shard) else None
        w_path, stats = _theta_path(w, ll, shard, rows, status if rows is not None else None)
        # (n, B_local) theta-weights after each observation; (n, 2): ESS, all finite
        event, status_host, host = None, None, None
        if rows is not None:
            pass
        elif stats.is_cuda:  # one small asynchronous copy into pinned memory + an event: the host later waits for THIS block only
            host = self._pinned(slot, stats)
            host.copy_(stats, non_blocking=True)
            if status is not None:
                status_host = self._pinned(("status", slot), status, rows=1)
                status_host.copy_(status, non_blocking=True)
            event = torch.cuda.Event()
            event.record()
        else:
            host = stats
        return dict(ys=ys, flags=flags, latest=latest, res=res, ll=ll, token=token, w_path=w_path, stats=stats, host=host, event=event,
                    w0=w, slot=slot, status_host=status_host, rows=rows)

    def _host_rows(self, slot: int, n: int):
        """The pipeline slot's statistics rows in host memory the device writes (``ops.HostRows``), or None where there is none to
        be had (then the statistics travel by a copy command and an event per block)."""
        from .. import ops

        bufs = self.__dict__.setdefault("_host_row_bufs", {})
        rows = bufs.get(slot)
        if rows is False:  # (asked once: no coherent host memory)
            return None
        if rows is None or rows.n < n:
            try:
                rows = ops.HostRows(max(n, self._block))
            except ops.L.PfAmdError:
                rows = False
            bufs[slot] = rows
        return rows or None

    def _verified(self, blk):
        """Waits for the block (its statistics and, for a column-cluster run, its status word).  A launch that gave up - the
        device was held by other work for seconds - is issued again on the per-step route from the same state on the same draws."""
        if blk["rows"] is not None:
            if blk["rows"].wait(0)[2] == 0:  # (every row carries the word: the first one's arrival is enough)
                return blk, False
        else:
            if blk["event"] is not None:
                blk["event"].synchronize()
            if blk["status_host"] is None or int(blk["status_host"][0]) == 0:
                return blk, False
        self.filter._cluster_gave_up(blk["res"].plan)
        again = self._issue_block(blk["ys"], blk["flags"], blk["latest"], blk["w0"], blk["slot"], token=blk["token"], per_step=True)
        return again, True

    def _pinned(self, slot, like: torch.Tensor, rows: Optional[int] = None) -> torch.Tensor:
        bufs = self.__dict__.setdefault("_pinned_bufs", {})
        key = (slot, like.dtype)
        buf = bufs.get(key)
        if buf is None or buf.shape[0] < like.shape[0]:
            shape = (max(like.shape[0], self._block), 2) if rows is None else (rows,)
            buf = bufs[key] = torch.empty(shape, dtype=like.dtype, pin_memory=True)
        return buf[: like.shape[0]]

    def _first_hit(self, blk) -> Optional[int]:
        """Waits for the block's statistics (only) and applies the reference's test to them in order."""
        thr = self._threshold * self.particles[0]
        if blk["rows"] is not None:  # row by row, each as soon as its workgroup is done - and no further than the first hit
            wait = blk["rows"].wait
            for q in range(blk["stats"].shape[0]):
                ess, finite, _ = wait(q)
                if ess < thr or not finite:
                    return q
            return None
        if blk["event"] is not None:
            blk["event"].synchronize()
        return next((q for q, (ess, finite) in enumerate(blk["host"].tolist()) if ess < thr or not finite), None)

    def _commit(self, blk, take: int, state: SMC2State) -> SMC2State:
        """The first ``take`` observations of a block become part of the algorithm state."""
        res, stats = blk["res"], blk["stats"]
        state.parsed.extend(blk["ys"][:take].unbind(0))
        state.w.copy_(blk["w_path"][take - 1])
        state.ess.extend(stats[:take, 0].unbind(0))
        state.stats = stats[take - 1]
        # the moves' moment rows as the run reported them - NOT a slice of the block result's series, whose window may be
        # bounded (record_moments=False / an int) and then no longer starts at the incoming state
        rows = getattr(res, "block_rows", None) or (res.filter_means[1:], res.filter_variance[1:])
        state.filter_state._extend_fused(rows[0], rows[1], res.loglikelihood, res.latest_state)
        state.current_iteration += take
        return state

    def _cut(self, blk, take: int, state: SMC2State):
        """The block again, cut after ``take`` observations, on the same draws."""
        res, ll, _ = self.filter.filter_block(blk["ys"][:take], blk["latest"], observed=blk["flags"][:take], replay=blk["token"])
        w_path, stats = _theta_path(state.w, ll, self.shard)
        return dict(blk, res=res, ll=ll, w_path=w_path, stats=stats)

    def fit(self, y: torch.Tensor, block: Optional[int] = None) -> SMC2State:
        """All observations of ``y``.  ``block`` (default: the constructor's) = how many observations the filters run
        ahead of the rejuvenation test; 1 = the reference's observation-by-observation loop (``step``).

        Reproducibility: for a fixed seed a fit is a function of ``(seed, block)``.  A block's moves draw from ONE Philox
        stream keyed by (the block's draw epoch, move index) - a cut block is replayed on exactly those draws - so another
        block length (or ``step()``, one epoch per move) is another, equally valid, Monte-Carlo run of the same algorithm;
        the SEQUENCE of decisions (when to rejuvenate, which moves are accepted given the numbers) is the reference's in
        every case (tests/test_inference_reference_gpu.py replays its event logs with ``block`` 1 and 16)."""
        state = self.initialize()
        k = self._block if block is None else max(1, int(block))
        flags = None
        if k > 1 and hasattr(self.filter, "filter_block") and isinstance(y, torch.Tensor) and y.is_floating_point():
            if y.is_cuda:  # which observations are not all-NaN: one device round trip for the whole series
                from .. import ops

                flags = ops.observed_flags(y if y.dtype in (torch.float32, torch.float64) else y.float()).cpu()
            else:
                flags = (~y.isnan().reshape(y.shape[0], -1).all(dim=1)).to(torch.uint8)
        state.attach_series(y, flags)
        t, total = 0, y.shape[0]
        pending, slot = None, 0  # the block in flight: issued, not yet decided
        while t < total:
            n = min(k, total - t)
            if flags is None or n < 2:
                state, t = self.step(y[t], state), t + 1
                continue
            blk = pending if pending is not None else self._issue_block(y[t:t + n], flags[t:t + n], state.filter_state.latest_state, state.w, slot)
            pending = None
            if blk is None:  # no fused block route: the observation-by-observation loop
                state, t = self.step(y[t], state), t + 1
                continue
            # speculate: the next block starts where this one ends - issued before the host looks at this one's statistics
            t_next = t + n
            n_next = min(k, total - t_next)
            draws_mark = getattr(self.filter, "_draws", None)  # (the filter's draw-epoch counter before the speculative issue)
            if n_next >= 2:
                slot ^= 1
                pending = self._issue_block(y[t_next:t_next + n_next], flags[t_next:t_next + n_next], blk["res"].latest_state,
                                            blk["w_path"][n - 1], slot)
            blk, reissued = self._verified(blk)
            if reissued and pending is not None:
                # (the speculative successor started from the state of a launch that gave up: dropped like one behind a
                # rejuvenation, its draw epoch handed back, and issued again from the re-issued block's state)
                if draws_mark is not None:
                    self.filter._draws = draws_mark
                pending = self._issue_block(y[t_next:t_next + n_next], flags[t_next:t_next + n_next], blk["res"].latest_state,
                                            blk["w_path"][n - 1], slot)
            hit = self._first_hit(blk)
            if hit is None:
                state, t = self._commit(blk, n, state), t_next
                continue
            if pending is not None and draws_mark is not None:
                # the speculative successor started from a state that never was: it is dropped, and the draw epoch it took is
                # handed back - the sequence of seeds a fit consumes does not depend on how often speculation failed
                self.filter._draws = draws_mark
            pending = None
            take = hit + 1
            if take < n:
                blk = self._cut(blk, take, state)
            state = self._commit(blk, take, state)
            t += take
            state = self._kernel.update(self.theta, self.filter, state, generator=self._gen)
        return state

    def posterior_mean(self, state: SMC2State) -> torch.Tensor:
        """Weighted mean of the stacked (constrained) parameters over ALL theta-particles."""
        vals, w = self.theta.stack_parameters(True), state.w
        if self.shard.collective:
            vals, w = self.shard.all_gather(vals), self.shard.all_gather(w)
        return theta_normalize(w) @ vals
