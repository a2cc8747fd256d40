"""
``ParticleFilter`` (``pyfilter/filters/particle/base.py:14-103,159-174``) with the MI355X fast path:

``batch_filter`` runs the whole time loop inside ``libpfamd.so`` (``pf_filter_run``: ONE fused HIP kernel per step,
no host synchronisation, ESS mask / NaN flags decided on the device) whenever the model is a built-in kind, the
proposal is ``Bootstrap`` / ``LinearGaussianObservations`` and the resampler is this package's ``systematic`` /
``multinomial``.  Anything else (user callables, custom resamplers, ``record_states=True``) takes the step-by-step
route through ``predict`` / ``correct`` - which still calls the HIP primitives for normalise / scan / search / gather /
moments and evaluates only the user's model callables with PyTorch-ROCm ops.
"""
import ctypes as C
from typing import Callable, Optional, Union

import torch

from ... import _lib as L
from ... import ops
from ...hints import HINTS
from ...resampling import multinomial, systematic
from ...timeseries import AffineEulerMaruyama, StateSpaceModel, TimeseriesState
from ...timeseries.models import pack_params
from ..base import BaseFilter
from ..result import FilterResult
from ..schedule import expand
from .proposals import Bootstrap, LinearGaussianObservations, Proposal
from .proposals.base import KernelContext
from .state import ParticleFilterCorrection, ParticleFilterPrediction

_DEFAULT_SEED = 2024
_GOLD = 0x9E3779B97F4A7C15  # odd 64-bit constant: consecutive draw epochs land on well-separated Philox keys
_M64 = 0xFFFFFFFFFFFFFFFF


class ParticleFilter(BaseFilter[ParticleFilterCorrection, ParticleFilterPrediction]):
    _FILTER_KIND = None  # PF_FILTER_*

    def __init__(
        self,
        model,
        particles: int,
        resampling: Callable[[torch.Tensor], torch.Tensor] = systematic,
        proposal: Union[str, Proposal] = None,
        ess_threshold=0.9,
        seed: Optional[int] = None,
        **kwargs,
    ):
        """
        Args:
            model: a ``StateSpaceModel`` or a builder ``context -> StateSpaceModel`` (see ``BaseFilter``).
            particles: number of particles N.
            resampling: ``systematic`` (default) or ``multinomial`` from :mod:`pyfilter_amd.resampling`, or any
                callable ``(w, normalized=False) -> LongTensor`` (then the step-by-step route is used).
            proposal: defaults to :class:`Bootstrap`.
            ess_threshold: relative ESS below which SISR resamples (``particle/base.py:42``).
            seed: seed of the in-kernel Philox generator (extension; the reference draws from torch's global RNG).
        """
        super().__init__(model, **kwargs)
        self._base_particles = torch.Size([particles])
        self._resample_threshold = ess_threshold * particles
        self._resampler = resampling
        self._proposal: Proposal = proposal if proposal is not None else Bootstrap()
        self._seed = _DEFAULT_SEED if seed is None else int(seed)
        self._ctx: Optional[KernelContext] = None
        self._z_tape = None
        self._u_tape = None
        self._z0 = None
        self._fused_plans = {}   # (shape, schedule) -> _FusedPlan (captured hipGraphs; a handful, evicted oldest first)
        self._single_plans = {}  # shape -> _SingleStepPlan (scratch of the online move; never evicted)
        self._single_last = None  # (the plan of the previous online move: checked field by field before the dictionary is asked)
        self._move_by_move = False  # testing knob: a fused run issued as its pieces (pf_filter_run(args, s, 1, 1), s = 0, 1, ..)
        self._draws = 0          # draw epoch: every new stream of random numbers (initial sample, fused run, online
                                 # move, step-by-step run) takes the next one - repeated calls are independent runs
        self._copies = 0
        self._defer_status_once = False  # the next batch_filter leaves a cluster run's verification to its caller
        self._per_step_once = False   # the next lean batch_filter takes the per-step route (see _batch_filter_lean)
        self._online_cluster = False  # set by a caller that verifies its online moves (SMC2.step): see _filter_fused_single
        self._watched_move = None     # (status word, redo) of the latest online move when it took the column-cluster kernel
        self._obs_cache = None   # (identity of y, host copy of its observed flags): re-filtering the same data costs no sync

    # ------------------------------------------------------------------------------------------------------------
    @property
    def particles(self) -> torch.Size:
        """``Size([N, *batch_shape])`` (particles first: ``particle/base.py:51-62``)."""
        return torch.Size([*self._base_particles, *self.batch_shape])

    @property
    def proposal(self) -> Proposal:
        return self._proposal

    def increase_particles(self, factor: int):
        self._base_particles = torch.Size([int(factor * self._base_particles[0])])
        self._resample_threshold *= factor
        self._ctx = None  # (tapes / scratch are sized by the particle count)

    def initialize_model(self, context):
        super().initialize_model(context)
        self._proposal.set_model(self._model)
        self._ctx = None  # the kernel context (kind, packed parameter rows) belongs to the model just replaced

    def set_tape(self, z: Optional[torch.Tensor] = None, u: Optional[torch.Tensor] = None, z0: Optional[torch.Tensor] = None):
        """Parity mode: inject the random draws instead of generating them with Philox.

        Args:
            z: standard normals per step in the reference's layout ``(T, N, [B], [D])``.
            u: resampling uniforms ``(T, B)`` (``(T, 1)`` / ``(T,)`` when unbatched).
            z0: standard normals ``(N, [B], [D])`` of the initial sample.
        """
        self._z_tape, self._u_tape, self._z0 = z, u, z0
        self._ctx = None

    # ------------------------------------------------------------------------------------------------------------
    @property
    def _batched(self) -> bool:
        return len(self.batch_shape) > 0

    @property
    def _has_event(self) -> bool:
        return self._model.hidden.n_dim > 0

    def _device_dtype(self):
        hidden = self._model.hidden
        dtype = next((p.dtype for p in hidden.parameters if p.is_floating_point()), torch.get_default_dtype())
        return hidden.device, dtype

    def _kernel_kind(self):
        return getattr(self._model, "kernel_kind", None)

    def _build_context(self, device, dtype) -> Optional[KernelContext]:
        kind = self._kernel_kind()
        if kind is None or device.type != "cuda":
            return None
        n = self._base_particles[0]
        b = self.batch_shape[0] if self._batched else 1
        params = pack_params(self._model, b, dtype, device)
        ctx = KernelContext(kind, params, self._seed, self._batched, self._has_event)
        ctx.params_signature = self._params_signature()  # as packed: any later in-place update re-packs
        ctx.init_host = None
        if self._z_tape is not None:
            z = self._z_tape.to(device=device, dtype=dtype)
            ctx.z_tape = z.reshape(z.shape[0], n, b, kind.dim).permute(0, 3, 2, 1).contiguous()  # (T, D, B, N)
        if self._u_tape is not None:
            ctx.u_tape = self._u_tape.to(device=device, dtype=dtype).reshape(self._u_tape.shape[0], b).contiguous()
        return ctx

    def _params_signature(self):
        """(identity, in-place version) of every model parameter tensor: cheap host-side change detection."""
        tensors = list(self._model.hidden.parameters) + list(self._model.parameters)
        init = getattr(self._model.hidden, "initial_parameters", None)
        if init is not None:
            tensors += list(init)
        return tuple((id(t), t._version) for t in tensors if isinstance(t, torch.Tensor))

    def _refresh_parameters(self):
        """The reference's model callables read the parameter tensors live, so in-place updates between calls (SMC^2 /
        PMMH ``exchange`` / ``resample`` of parameters) take effect at once.  The kernels read a packed copy: re-pack it
        (into the same buffer - captured graphs keep their pointer) whenever a parameter tensor was modified in place."""
        ctx = self._ctx
        if ctx is None:
            return
        sig = self._params_signature()
        if ctx.params_signature != sig:
            b = self.batch_shape[0] if self._batched else 1
            ctx.params.copy_(pack_params(self._model, b, ctx.params.dtype, ctx.params.device))
            ctx.params_signature = sig
            ctx.init_host = None

    def _ensure_context(self):
        """The kernel context of the current model / tapes (built once, parameters re-packed when they change)."""
        if self._ctx is None:
            self._proposal.set_model(self.ssm)
            self._ctx = self._build_context(*self._device_dtype())
            self._proposal._set_context(self._ctx)
        self._refresh_parameters()
        return self._ctx

    @property
    def _run_seed(self) -> int:
        """Seed of the current step-by-step run (set by ``initialize``)."""
        return self._ctx.seed if self._ctx is not None else (self._seed + _GOLD * self._draws) & _M64

    def _next_draw_seed(self) -> int:
        """Seed of the next independent stream of Philox draws."""
        self._draws += 1
        seed = (self._seed + _GOLD * self._draws) & _M64
        if self._ctx is not None:
            self._ctx.seed = seed  # the step-by-step kernels of this run key their draws by (seed, time index)
        return seed

    def initialize(self) -> ParticleFilterCorrection:
        assert self._model is not None, "Model has not been initialized!"
        device, dtype = self._device_dtype()
        ctx = self._ensure_context()
        seed = self._next_draw_seed()

        n = self._base_particles[0]
        b = self.batch_shape[0] if self._batched else 1
        hidden = self._model.hidden
        if ctx is not None and hasattr(hidden, "init_mean"):
            d = ctx.kind.dim
            z0 = None
            if self._z0 is not None:
                z0 = ops.to_soa(self._z0.to(device=device, dtype=dtype), self._batched, self._has_event)
            im, isd = hidden.init_mean, hidden.init_scale
            if im.numel() in (1, d) and isd.numel() in (1, d):
                # host copies of the initial mean / scale, taken once per parameter change: any per-call transfer of these
                # few numbers (host -> device for CPU-resident constants, device -> host for device ones) would make the
                # host wait for the whole queue of the previous run
                if ctx.init_host is None:
                    ctx.init_host = (im.reshape(-1).expand(d).to(torch.float64).tolist(), isd.reshape(-1).expand(d).to(torch.float64).tolist())
                soa = ops.initial_sample_soa(ctx.init_host[0], ctx.init_host[1], n, b, d, dtype, device, seed, z0)
            else:
                im, isd = im.to(device=device, dtype=dtype), isd.to(device=device, dtype=dtype)
                # per-filter initial parameters (theta on the batch dim): m + s z inside the sampling kernel - the parameters
                # are read through the strides of their broadcast views
                from ...timeseries.models import _expand
                soa = ops.initial_sample_cols(_expand(im, b, (d,), dtype, device), _expand(isd, b, (d,), dtype, device), n, b, d, seed, z0)
            x = TimeseriesState(0, ops.from_soa(soa, self._batched, self._has_event), hidden.event_shape)
        else:
            x = hidden.initial_sample(self.particles)
            if x.value.device != device:
                x = x.copy(values=x.value.to(device))

        w_cols = torch.zeros((b, n), device=device, dtype=x.value.dtype)
        weights = ops.from_cols(w_cols, self._batched)
        prev_inds = torch.arange(n, device=device)
        if self._batched:
            prev_inds = prev_inds.unsqueeze(-1).expand(self.particles)
        ll = torch.zeros(self.batch_shape, device=device, dtype=x.value.dtype)
        return ParticleFilterCorrection(x, weights, ll, prev_inds)

    def _propagate_only(self, prediction):
        return prediction.create_state_from_prediction(self._model, propagate=self._proposal._propagate)

    def _copy_seed(self) -> int:
        """A copy draws its own random numbers (the reference's copies share torch's global generator, i.e. never repeat
        the original's draws); with tapes set the seed is irrelevant."""
        self._copies += 1
        return (self._seed ^ (_GOLD * (self._copies + 0x51ED27))) & _M64

    def copy(self):
        """NB: like the reference (``particle/base.py:165``) the *absolute* threshold is handed to the copy's
        ``ess_threshold`` - a copied SISR therefore effectively always resamples."""
        res = type(self)(
            model=self._model_builder,
            particles=self._base_particles[0],
            resampling=self._resampler,
            proposal=self._proposal.copy(),
            ess_threshold=self._resample_threshold,
            seed=self._copy_seed(),
            record_states=self.record_states,
            record_moments=self.record_moments,
            nan_strategy=self._nan_strategy,
            record_intermediary_states=self._record_intermediary,
        )
        res.set_batch_shape(self.batch_shape)
        return res

    # ------------------------------------------------------------------------------------------------------------
    # resampling helpers used by SISR / APF on the step-by-step route
    # ------------------------------------------------------------------------------------------------------------
    def _resampler_kind(self) -> Optional[int]:
        if self._resampler is systematic:
            return L.RESAMPLE_SYSTEMATIC
        if self._resampler is multinomial:
            return L.RESAMPLE_MULTINOMIAL
        return None

    def _ancestors_of(self, log_weights: torch.Tensor, time_index: int) -> torch.Tensor:
        """Ancestors (int64, the shape of ``log_weights``) drawn for every filter from unnormalised log-weights on the
        step-by-step route: the library's systematic / multinomial kernels, or the user's resampler as a callable."""
        kind = self._resampler_kind()
        if kind is None:
            return self._resampler(log_weights)
        cols = ops.to_cols(log_weights)
        if kind == L.RESAMPLE_SYSTEMATIC:
            picked = ops.systematic_cols(cols, self._uniforms(time_index, cols.shape[0], cols), normalized=False)
        else:
            picked = ops.multinomial_cols(ops.normalize_cols(cols, want_w=True)[0], self._run_seed, step=time_index)
        return ops.from_cols(picked, log_weights.dim() > 1).long()

    def _uniforms(self, step: int, b: int, like: torch.Tensor) -> torch.Tensor:
        """The step's systematic offsets on the step-by-step route: the tape, or a device generator keyed by the run's
        draw seed (so a rebuilt filter with the same seed repeats them and a second run does not)."""
        if self._ctx is not None and self._ctx.u_tape is not None:
            return self._ctx.u_tape[step]
        key = (self._run_seed, like.device)
        gen = getattr(self, "_u_gen", None)
        if gen is None or gen[0] != key:
            g = torch.Generator(device=like.device)
            g.manual_seed(self._run_seed & 0x7FFFFFFFFFFFFFFF)
            gen = self._u_gen = (key, g)
        return torch.empty(b, device=like.device, dtype=like.dtype).uniform_(generator=gen[1])

    # ------------------------------------------------------------------------------------------------------------
    # the fused loop
    # ------------------------------------------------------------------------------------------------------------
    def _fused_capable(self, device) -> bool:
        return (
            device.type == "cuda"
            and self._FILTER_KIND is not None
            and self._kernel_kind() is not None
            and type(self._proposal) in (Bootstrap, LinearGaussianObservations)
            and not self._proposal._custom_pre_weight
            and self._resampler_kind() is not None
            and (hasattr(self._model.hidden, "init_mean") or self._kernel_kind().is_user)
        )

    def filter(self, y: torch.Tensor, correction: ParticleFilterCorrection, result: FilterResult = None):
        """One filter move (``filters/base.py:188-221``).  Built-in models take the fused single-step path; anything
        else (user callables, custom resamplers, ``observe_every_step > 1``, recorded intermediary states) the
        reference's predict / correct sequence over the stand-alone kernels."""
        x = correction.timeseries_state.value
        if (not isinstance(y, torch.Tensor) or not self._fused_capable(x.device) or int(self._model.observe_every_step) != 1
                or self._record_intermediary or not HINTS.fused_step):
            return super().filter(y, correction, result=result)
        if result is None:
            return self._filter_fused_single(y, correction)
        # (the run adds the move's log-likelihood to the result's running total itself - ``pf_filter_args.ll_total`` IS that
        # tensor - when it can be handed over as it is: one elementwise launch per online move less)
        new = self._filter_fused_single(y, correction, ll_into=result._loglikelihood)
        # (a watched move - ``_online_cluster`` - leaves the accumulation to its caller's pf_theta_step: ``_watched_move``)
        result.append(new, _ll_accumulated=self._ll_accumulated or self._watched_move is not None)
        return new

    def _filter_fused_single(self, y: torch.Tensor, state: ParticleFilterCorrection, ll_into: torch.Tensor = None) -> ParticleFilterCorrection:
        # (the host's share of an online move is what SMC2.step() / a driver's loop over filter() run at - every torch call here
        # costs 1 - 1.5 us: the buffers of a state this method produced are remembered with it, a plan keeps what does not
        # change between moves, pointers into the move's statistics block are computed, not sliced)
        ctx = self._ensure_context()
        kind = ctx.kind
        ts_in = state.timeseries_state
        batched, has_event = self._batched, self._has_event
        soa = getattr(state, "_soa", None)
        if soa is not None and soa[0] is ts_in.value and soa[1] is state["_w"]:
            x_in, lw_in = soa[2], soa[3]  # (the kernels' own buffers of a state nobody replaced since)
        else:
            x_in = ops.to_soa(ts_in.value, batched, has_event)   # views of library buffers: no copies
            lw_in = ops.to_cols(state.weights)
        device, dtype = x_in.device, x_in.dtype
        d, b, n = x_in.shape
        o = kind.obs_dim
        y_dev = (y if (y.dtype == dtype and y.device == device) else y.to(device=device, dtype=dtype)).reshape(1, -1, o)
        if not y_dev.is_contiguous():
            y_dev = y_dev.contiguous()
        rows = y_dev.shape[1]
        if rows != 1 and rows != b:
            raise L.PfAmdError(f"observation of shape {tuple(y.shape)} does not broadcast against batch {b}")
        rs_kind, thr = self._resampler_kind(), float(self._resample_threshold)
        plan = self._single_last
        if (plan is None or plan.n != n or plan.b != b or plan.d != d or plan.o != o or plan.rows != rows or plan.dtype != dtype
                or plan.device != device or plan.rs_kind != rs_kind or plan.thr != thr or plan.kind is not kind):
            key = (n, b, d, o, rows, dtype, device, self._FILTER_KIND, self._proposal._KERNEL_PROPOSAL, rs_kind, thr)
            plan = self._single_plans.get(key)
            if plan is None:
                plan = self._single_plans[key] = _SingleStepPlan(self, kind, n, b, d, o, rows, dtype, device)
            plan.rs_kind, plan.thr, plan.kind = rs_kind, thr, kind
            self._single_last = plan
        t_start = int(ts_in.time_index)
        # all-NaN observation -> propagate only (filters/base.py:212).  The reference branches on the host; here the flag
        # is derived on the device by the run itself (neither flag array passed), so consecutive filter() calls never
        # wait for the GPU

        apf = self._FILTER_KIND == L.FILTER_APF
        x_out, lw_out = torch.empty_like(x_in), torch.empty_like(lw_in)
        if apf:
            anc = torch.empty((b, n), device=device, dtype=torch.int32)  # every APF step writes its ancestors
        else:  # SISR keeps the previous ancestors when it does not resample (sisr.py:25-26)
            own = state._anc32 is not None  # the kernels' buffer of the incoming state: copied, it stays that state's
            anc = state.ancestors32().clone() if own else state.ancestors32().reshape(b, n).contiguous()
        # the move's statistics block, zeroed: means (2, B, D) | variances (2, B, D) | ll (B) | total (B)
        mean_new, var_new, ll_new, stats_ptr = plan.zeroed_stats(batched)
        es = plan.elem_size
        a = plan.args
        # An online move is not verified by anybody who waits for it - the caller may never look at the device again before the
        # next move - so it takes the column-cluster kernel (whose launches can give up, hints.py) only on behalf of a caller
        # that reads the move's status word with the next thing it waits for: SMC2.step() (``_online_cluster``; the word travels
        # through pf_theta_step into the host slot it polls).  Everybody else's moves of that size stay on the per-step route.
        verified = self._online_cluster and HINTS.kernel_route() == 3 and HINTS.cluster_takes(n, b, rs_kind == L.RESAMPLE_SYSTEMATIC)
        stream = L.stream_ptr()
        hk = (HINTS.key(), verified, stream)  # (the stream: a resumed piece relies on stream order behind the previous one)
        if plan.hints_key != hk or a.hints.prepare_next:
            HINTS.fill(a)
            if a.hints.route == 3 and not verified:
                a.hints.route = 0  # PF_ROUTE_AUTO
            plan.hints_key = hk
        if verified:
            ll_into = None  # (the running total is accumulated by pf_theta_step, under the same status word: SMC2State.append)
        a.model.params = ctx.params.data_ptr()
        planes = None
        if kind.is_user:
            # The user's mean_scale callable, ONCE per step, on the incoming particles (torch ops on the device): it acts per
            # particle, so loc(x[anc]) = loc(x)[anc] - the kernels gather these planes at the ancestors instead of
            # evaluating a built-in closed form.  Everything else of the step stays in the fused kernels.
            loc, scale, a.user_dt = self._user_mean_scale(self._model.hidden, ts_in)
            full = ts_in.value.shape
            percol = self._scale_per_column(scale, full, dtype)
            planes = (ops.to_soa(loc.to(dtype).expand(full), batched, has_event).contiguous(),
                      percol if percol is not None else ops.to_soa(scale.to(dtype).expand(full), batched, has_event).contiguous())
            a.user_loc, a.user_scale = planes[0].data_ptr(), planes[1].data_ptr()
            a.user_scale_per_column = 0 if percol is None else 1
        # ---- the resume token (include/pf_amd.h: a run issued in pieces on ONE argument block) --------------------------------------
        # A move on the per-step route is piece m of a run whose pieces the workspace connects: when the incoming state is exactly what
        # this plan's previous move wrote - the same tensors, untouched since (their in-place version counters) - the move is issued
        # as piece m + 1: no launch that clears the per-filter records, and for SISR no launch that re-reduces the incoming state (its
        # partials and local scans are what the previous step kernel left: pf_run_hints.resume; an APF's first stage weighs with the
        # NEW observation, so its reduce pass stays).  Rows are addressed relative to piece m: the pointers below are offset so that
        # row m is this move's.  Anything else - a state somebody replaced or edited, another plan's, a rejuvenated filter set -
        # starts a fresh run at piece 0.
        m = 0
        per_step_route = (not verified) and (HINTS.route == 1 or (a.hints.route == 0 and n > (HINTS.column_max_n or 2048)))
        chain = plan.chain
        if (per_step_route and chain is not None and chain[1] is x_in and chain[2] is lw_in and chain[3] == x_in._version
                and chain[4] == lw_in._version and chain[5] == hk):
            m = chain[0] + 1
        plan.chain = None
        resume = 1 if (m > 0 and not apf) else 0
        if a.hints.resume != resume:
            a.hints.resume = resume
        a.y, a.y_rows = y_dev.data_ptr() - m * rows * o * es, rows
        a.observed, a.observed_dev, a.step_counter = None, None, None  # (the block route shares this argument block)
        a.seed = self._next_draw_seed()  # fresh Philox draws per move
        a.x[m & 1], a.x[(m + 1) & 1] = x_in.data_ptr(), x_out.data_ptr()
        a.logw[m & 1], a.logw[(m + 1) & 1] = lw_in.data_ptr(), lw_out.data_ptr()
        a.anc = anc.data_ptr()
        db = d * b * es
        a.means, a.vars = stats_ptr - m * db, stats_ptr + 2 * db - m * db
        a.ll_steps = stats_ptr + 4 * db - m * b * es
        self._ll_accumulated = (ll_into is not None and ll_into.device == device and ll_into.dtype == dtype and ll_into.numel() == b
                                and ll_into.is_contiguous())
        a.ll_total = ll_into.data_ptr() if self._ll_accumulated else stats_ptr + 4 * db + b * es
        z_tape = u_tape = None
        if ctx.z_tape is not None:
            z_tape = ctx.z_tape[t_start:t_start + 1].contiguous()
            assert z_tape.shape[0] == 1, "z tape shorter than the number of steps"
        if ctx.u_tape is not None:
            u_tape = ctx.u_tape[t_start:t_start + 1].contiguous()
        # no uniform tape: every workgroup draws its column's u (Philox)
        a.z_tape = None if z_tape is None else z_tape.data_ptr() - m * d * b * n * es
        a.u_tape = None if u_tape is None else u_tape.data_ptr() - m * b * es
        plan.generation = g_ = plan.generation % 0xFFFFF + 1  # (numbered cluster launches: tagged records, no clearing launch)
        a.hints.cluster_generation = g_
        L.check(plan.run(plan.args_ref, m, 1, 1, stream), "pf_filter_run")
        if per_step_route:
            plan.chain = (m, x_out, lw_out, x_out._version, lw_out._version, hk)
        self._last_run = dict(plan=plan, z=z_tape, u=u_tape, ws=plan.ws, seed_eff=a.seed, piece=m,
                              keep=(x_in, lw_in, y_dev, ctx.params, planes))
        self._watched_move = None
        if verified:
            pool_row = plan._pool[0][plan._pool_next - 1]

            def redo():
                """The same move again on the per-step route, into the same tensors (the argument block still describes it)."""
                self._cluster_gave_up(plan)
                if not apf:
                    anc.copy_(state.ancestors32().reshape(b, n))
                pool_row[4 * d + 1].zero_()  # (the move's own total: accumulated, not written)
                a.hints.route, plan.hints_key = 1, None  # PF_ROUTE_PER_STEP (the next move re-writes the hints)
                L.check(plan.run(plan.args_ref, 0, 1, 1, L.stream_ptr()), "pf_filter_run")

            self._watched_move = (plan.status, redo)

        x_view, w_view = plan.state_views(x_out, lw_out, batched, has_event)
        new = ParticleFilterCorrection(TimeseriesState(t_start + 1, x_view, self._model.hidden.event_shape), w_view, ll_new, None,
                                       _moments=(mean_new, var_new), _anc32=(anc, batched))
        new._soa = (x_view, w_view, x_out, lw_out)
        return new

    def batch_filter(self, y, bar=True, init_state=None) -> FilterResult:
        """``self._defer_status_once`` (set by callers inside this package that never wait for a run by itself - PMMH moves): a run
        that took the column-cluster kernel is NOT verified before it is handed back; ``result._cluster_watch = (status word,
        plan)`` lets the caller look where it next waits for the device (a launch that gave up leaves NaN log-likelihoods: a
        rejected proposal)."""
        assert self._model is not None, "Model has not been initialized!"
        _defer_status, self._defer_status_once = self._defer_status_once, False
        device, _ = self._device_dtype()
        if (not self._fused_capable(device) or not isinstance(y, torch.Tensor)
                or not HINTS.fused_batch
                or (self._kernel_kind().is_user and FilterResult.states_kept(self.record_states) != 1)):
            # (a user-defined affine process with recorded states: the driver's loop over fused single steps)
            return super().batch_filter(y, bar=bar, init_state=init_state)
        if self._single_launch_run(y):
            return self._batch_filter_lean(y, init_state, defer_status=_defer_status)
        return self._batch_filter_fused(y, init_state)

    def _single_launch_run(self, y) -> bool:
        """A run the library issues as ONE launch of the column-persistent kernel (filters of <= 2 048 particles) - or one or two
        of the column-cluster kernel (2 049 .. 16 384 particles, ``HINTS.cluster_takes``) - with nothing
        recorded but the moments: there is no launch sequence for a hipGraph to replay, so the persistent plan of the general
        driver (its buffers, staging copies and per-shape cache - PMMH re-filters a data set of another length at every
        rejuvenation) only costs host time."""
        return (y.shape[0] > 0 and FilterResult.states_kept(self.record_states) == 1 and not getattr(self, "_time_kernels", False)
                and not self._move_by_move and not self._kernel_kind().is_user and int(self._model.observe_every_step) == 1
                and (HINTS.direct or (HINTS.route != 1 and self._base_particles[0] <= (HINTS.column_max_n or 2048))
                     or HINTS.cluster_takes(self._base_particles[0], self.batch_shape[0] if self._batched else 1,
                                            self._resampler_kind() == L.RESAMPLE_SYSTEMATIC))
                and self._ctx_tapes_none())

    def _batch_filter_lean(self, y: torch.Tensor, init_state=None, defer_status: bool = False) -> FilterResult:
        state = init_state if init_state is not None else self.initialize()
        result = FilterResult(state, self.record_states, self.record_moments, _defer_moments=True)
        per_step, self._per_step_once = self._per_step_once, False  # (a caller repeating a run whose cluster launch gave up)
        blk, _, _ = self._filter_block_lean(y, state, None, None, host_u=True, defer_status=defer_status, per_step=per_step)
        result._cluster_watch = (blk.status, blk.plan) if blk.status is not None else None
        result._extend_fused(blk.filter_means, blk.filter_variance, blk.loglikelihood, blk.latest_state)
        self._last_run["rows"] = (blk.filter_means, blk.filter_variance)
        return result

    def _observed_flags(self, y: torch.Tensor, y_dev: torch.Tensor) -> torch.Tensor:
        """Host copy of "observation k carries information" (not all-NaN), one byte per observation.  The launch loop
        reads the flags on the host (they select the kernel variant of every step), which costs one device round trip
        per *new* data set: PMMH / SMC^2 re-filter the same ``y`` over and over, and for that the flags are remembered
        by the tensor object and its in-place version."""
        cached = self._obs_cache  # (the tensor object itself - an address alone could be a recycled allocation -, version, flags)
        if cached is not None and cached[0] is y and cached[1] == y._version:
            return cached[2]
        flags = ops.observed_flags(y_dev).cpu()
        self._obs_cache = (y, y._version, flags)
        return flags

    def filter_block(self, y: torch.Tensor, state: ParticleFilterCorrection, observed: Optional[torch.Tensor] = None,
                     replay=None, per_step: bool = False, defer_status: bool = False):
        """``len(y)`` consecutive moves from ``state`` as ONE fused run - what a caller that decides something on the host
        after every observation (SMC^2: rejuvenate or not, ``smc2.py:59-62``) uses to look ahead: it runs a block, reads
        the per-move log-likelihood increments once, and if its decision fell at move ``j`` inside the block asks for the
        block again cut after that move - ``replay = token`` of the first run - which repeats exactly the same draws, so
        the state it gets is the one the increments it already used belong to.

        Returns ``(result, ll_steps, token)``: the run's ``FilterResult`` (moment rows incl. the incoming state's, total
        log-likelihood, final state), the increments ``(len(y), *batch_shape)`` and the token for a replay.  ``observed``:
        optional host flags (uint8, one per observation) when the caller already knows which observations are not
        all-NaN (saves the device round trip per call).  None when the fused route does not apply.

        A block that took the column-cluster kernel is verified before it is handed back (its status word read, the block
        re-issued on the per-step route had the launch given up - ``_filter_block_lean``).  ``defer_status``: the caller reads
        the word itself - ``result.status``, a device int32 tensor or None, valid in stream order after the block - and on a
        non-zero value calls ``_cluster_gave_up`` and asks again with ``replay = token, per_step = True``."""
        x = state.timeseries_state.value
        if (not self._fused_capable(x.device) or int(self._model.observe_every_step) != 1 or self._record_intermediary
                or self._kernel_kind().is_user):
            return None
        if (FilterResult.states_kept(self.record_states) == 1 and not getattr(self, "_time_kernels", False)
                and not self._move_by_move and self._ctx_tapes_none()):
            return self._filter_block_lean(y, state, observed, replay, per_step=per_step, defer_status=defer_status)
        res = self._batch_filter_fused(y, state._restarted(), observed=observed, replay=replay)
        res.status = None
        run = self._last_run
        # the moves' own moment rows (row 0 = the incoming state) - copies: a cached plan's buffers are rewritten by its next run
        res.block_rows = (run["rows"][0][1:].clone(), run["rows"][1][1:].clone())
        ll = run["ll_steps"]
        u = run["u"]  # (a cached plan's buffer, redrawn by the next run: the token keeps a copy)
        return res, (ll if self._batched else ll[:, 0]), (run["seed_eff"], None if u is None else u.clone())

    @staticmethod
    def _user_mean_scale(hidden, ts):
        """``(loc, scale, dt)`` of a user-defined affine process for one fused move: the callable's one-step mean and scale
        with ``dt = 0``, or - an exact ``AffineEulerMaruyama`` with a plain-number ``dt`` - its DRIFT, scale and ``dt``: the
        kernels then form ``x + f dt`` at the parent themselves (``pf_filter_args.user_dt``)."""
        if type(hidden) is AffineEulerMaruyama:
            fg = hidden.drift_scale(ts)
            if fg is not None:
                return fg
        loc, scale = hidden.mean_scale(ts)
        return loc, scale, 0.0

    def _scale_per_column(self, scale: torch.Tensor, full, dtype) -> Optional[torch.Tensor]:
        """The user's transition scale as a ``(D, B)`` array when it does not vary along the particle dimension (a broadcast
        scalar / per-filter / per-component value: stride 0 along dim 0 of the broadcast ``mean_scale`` returns) - what
        ``pf_filter_args.user_scale_per_column`` takes instead of a ``(D, B, N)`` plane filled per move.  None otherwise."""
        if scale.dim() != len(full) or scale.shape != full or scale.stride(0) != 0:
            return None
        # (a callable typically hands back the same parameter tensor move after move: the small array is built once per
        # storage / in-place version, not per move - four tiny launches a host-bound small filter would feel)
        key = (scale.data_ptr(), scale._version, tuple(scale.stride()), dtype)
        cached = getattr(self, "_percol_cache", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        out = self._scale_rows(scale, dtype)
        # (the view is kept with the entry: while it lives its storage cannot be handed to another tensor, so an equal
        # address + version really is the same data - a callable that computes a fresh scale per move misses, as it must)
        self._percol_cache = (key, out, scale)
        return out

    def _scale_rows(self, scale: torch.Tensor, dtype) -> torch.Tensor:
        row = scale[0].to(dtype)                                   # ([B], [D])
        if not self._batched:
            row = row.unsqueeze(0)                                 # (1, [D])
        row = row.reshape(row.shape[0], -1)                        # (B, D)
        return row.t().contiguous()                                # (D, B)

    def _ctx_tapes_none(self) -> bool:
        ctx = self._ensure_context()
        return ctx.z_tape is None and ctx.u_tape is None

    def online_run(self, result: FilterResult):
        """The fast driver of an observation-by-observation loop over ``result`` (``_OnlineRun``), or None where it does not apply."""
        return _OnlineRun(self, result) if _OnlineRun.applies(self, result) else None

    def _cluster_gave_up(self, plan):
        """A column-cluster launch of ``plan`` reported that it could not make progress: the word is cleared for the next
        run and the event is announced once per filter object (the caller re-issues the piece on the per-step route)."""
        plan.status.zero_()
        self.cluster_fallbacks = getattr(self, "cluster_fallbacks", 0) + 1
        if self.cluster_fallbacks == 1:
            import warnings

            warnings.warn("pyfilter_amd: a column-cluster launch gave up waiting for its sibling workgroups (the device was held "
                          "by other work); the piece is re-issued on the per-step route - same draws, same results")

    def _filter_block_lean(self, y: torch.Tensor, state: ParticleFilterCorrection, observed, replay, host_u: bool = False,
                           per_step: bool = False, defer_status: bool = False):
        """``filter_block`` for the caller it exists for - SMC^2, which issues a block of ~16 moves per host decision, a few
        dozen blocks per fit: at 1 000 theta x 400 particles such a block is 90 us of kernel time, and the general fused
        driver (persistent plan, staging copies in and out, a ``FilterResult`` with its moment log per call: ~25 small
        torch ops, 0.25 ms of host time) was what a fit consisted of.  Here the run writes straight into the tensors that
        become the new state and the block's rows (four allocations), reads the incoming state through one packed copy,
        takes parameters / observations / flags where they are, draws its systematic offsets in the kernels (keyed by seed
        and move, so a cut replay repeats them) - ten device operations and one ``pf_filter_run`` per block."""
        ctx = self._ctx
        kind = ctx.kind
        ts_in = state.timeseries_state
        x_in = ops.to_soa(ts_in.value, self._batched, self._has_event)
        lw_in = ops.to_cols(state.weights)
        device, dtype = x_in.device, x_in.dtype
        d, b, n = x_in.shape
        o = kind.obs_dim
        steps = y.shape[0]
        y_dev = y.to(device=device, dtype=dtype).reshape(steps, -1, o)
        if not y_dev.is_contiguous():
            y_dev = y_dev.contiguous()
        rows = y_dev.shape[1]
        if rows not in (1, b):
            raise L.PfAmdError(f"observations of shape {tuple(y.shape)} do not broadcast against batch {b}")
        flags = (observed if observed is not None else self._observed_flags(y, y_dev)).contiguous()
        key = (n, b, d, o, rows, dtype, device, self._FILTER_KIND, self._proposal._KERNEL_PROPOSAL,
               self._resampler_kind(), float(self._resample_threshold))
        plan = self._single_plans.get(key)
        if plan is None:
            plan = self._single_plans[key] = _SingleStepPlan(self, kind, n, b, d, o, rows, dtype, device)
        plan.chain = None  # (the block's run rewrites the workspace an online move's resume token points into)
        if plan.xl is None:  # the run's other state slot (the kernels alternate between two)
            plan.xl = torch.empty((d + 1, b, n), device=device, dtype=dtype)
        t_start = int(ts_in.time_index)

        # what the run writes: the new state (particles + log-weights in one allocation, ancestors), the rows, the increments
        xl_out = torch.empty((d + 1, b, n), device=device, dtype=dtype)
        anc = torch.empty((b, n), device=device, dtype=torch.int32)
        rows_buf = torch.empty((2, steps + 1, b, d), device=device, dtype=dtype)
        ll = torch.zeros((steps + 1, b), device=device, dtype=dtype)  # per move | the run's total (accumulated in place)
        # the incoming state goes to slot 0; the final state lands in slot steps & 1 - which must be the fresh allocation
        first = xl_out if steps % 2 == 0 else plan.xl
        other = plan.xl if steps % 2 == 0 else xl_out
        xl_in = getattr(state, "_xl", None)
        packed = (xl_in is not None and xl_in.shape == first.shape and xl_in.dtype == dtype and xl_in.data_ptr() == x_in.data_ptr()
                  and xl_in[d].data_ptr() == lw_in.data_ptr())

        def load_state():
            if packed:
                first.copy_(xl_in)  # (a state this route produced and nobody replaced since: particles and log-weights sit in one buffer)
            else:
                first[:d].copy_(x_in)
                first[d].copy_(lw_in)
            if self._FILTER_KIND != L.FILTER_APF:  # (an APF names new ancestors at every move: it never reads the incoming ones)
                anc.copy_(state.ancestors32().reshape(b, n))

        load_state()
        a = plan.args
        HINTS.fill(a)
        if per_step:
            a.hints.route = 1  # PF_ROUTE_PER_STEP: the re-issue of a piece whose column-cluster launch gave up
        a.model.params = ctx.params.data_ptr()
        a.y, a.y_rows = y_dev.data_ptr(), rows
        a.observed, a.observed_dev = flags.data_ptr(), None
        seed_eff = self._next_draw_seed() if replay is None else replay[0]
        a.seed, a.step_counter = seed_eff, None
        u_tape = replay[1][:steps].contiguous() if (replay is not None and replay[1] is not None) else None
        if host_u and u_tape is None and self._resampler_kind() == L.RESAMPLE_SYSTEMATIC:
            # ``batch_filter``: the offsets of the general driver (a device generator seeded by the run's seed), so that a
            # run draws the same numbers whichever driver - and kernel route - carries it
            if plan.u_gen is None:
                plan.u_gen = torch.Generator(device=device)
            plan.u_gen.manual_seed(seed_eff & 0x7FFFFFFFFFFFFFFF)
            u_tape = torch.empty((steps, b), device=device, dtype=dtype).uniform_(generator=plan.u_gen)
        a.z_tape, a.u_tape = None, L.ptr(u_tape)
        a.x[0], a.x[1] = first.data_ptr(), other.data_ptr()
        a.logw[0], a.logw[1] = first[d].data_ptr(), other[d].data_ptr()
        a.anc = anc.data_ptr()
        a.means, a.vars = rows_buf[0].data_ptr(), rows_buf[1].data_ptr()
        a.ll_steps, a.ll_total = ll.data_ptr(), ll[steps].data_ptr()
        plan.generation = a.hints.cluster_generation = plan.generation % 0xFFFFF + 1
        L.check(L.load().pf_filter_run(C.byref(a), 0, steps, 1, L.stream_ptr()), "pf_filter_run")
        # A column-cluster launch reports through the plan's status word when it could not make progress (include/pf_amd.h:
        # PF_ROUTE_CLUSTER).  ``batch_filter`` looks at it here (one small device -> host read per run); a caller that pipelines
        # blocks (SMC2.fit) takes the word with the block (``defer_status``), reads it with the block's statistics and asks for
        # the block again with ``per_step = True`` - a replay on the same draws, so the numbers are the one-piece run's.
        watched = (not per_step) and HINTS.cluster_takes(n, b, self._resampler_kind() == L.RESAMPLE_SYSTEMATIC)
        if watched and not defer_status and int(plan.status.item()) != 0:
            self._cluster_gave_up(plan)
            load_state()
            ll.zero_()
            a.hints.route = 1
            L.check(L.load().pf_filter_run(C.byref(a), 0, steps, 1, L.stream_ptr()), "pf_filter_run")
        self._last_run = dict(plan=plan, z=None, u=u_tape, ws=plan.ws, seed_eff=seed_eff, ll_steps=ll[:steps],
                              keep=(x_in, lw_in, y_dev, flags, ctx.params))

        md = (lambda t: t) if self._batched else (lambda t: t[:, 0])
        means, variances = md(rows_buf[0]), md(rows_buf[1])
        ll_steps = ll[:steps] if self._batched else ll[:steps, 0]
        last = ParticleFilterCorrection(
            TimeseriesState(t_start + steps, ops.from_soa(xl_out[:d], self._batched, self._has_event), self._model.hidden.event_shape),
            ops.from_cols(xl_out[d], self._batched), ll_steps[steps - 1], None,
            _moments=(means[steps], variances[steps]), _anc32=(anc, self._batched),
        )
        last._xl = xl_out
        res = _BlockResult(means, variances, ll[steps] if self._batched else ll[steps, 0], last)
        res.status = plan.status if (watched and defer_status) else None  # (device int32 word, sticky: see above)
        res.plan = plan
        return res, ll_steps, (seed_eff, None)

    def _batch_filter_fused(self, y: torch.Tensor, init_state=None, observed=None, replay=None) -> FilterResult:
        state = init_state if init_state is not None else self.initialize()
        ctx = self._ensure_context()
        kind = ctx.kind
        x0 = state.timeseries_state.value
        device, dtype = x0.device, x0.dtype
        n = self._base_particles[0]
        b = self.batch_shape[0] if self._batched else 1
        d, o = kind.dim, kind.obs_dim

        # row 0 of the moment series comes from the kernels too (the bookkeeper of the first launch reduces the incoming
        # state), so the initial state's moments are not reduced a second time on the way in
        result = FilterResult(state, self.record_states, self.record_moments, _defer_moments=True)
        t_obs = y.shape[0]
        if t_obs == 0:
            result._moments.append(state.get_mean(), state.get_variance(), self._batched)
            return result

        # ---- the run's moves (filters/schedule.py): observe_every_step > 1 inserts propagate-only sub-steps ----------
        t_start = int(state.timeseries_state.time_index)
        sched = expand(t_start, t_obs, int(self._model.observe_every_step))
        steps = sched.moves
        y_dev = y.to(device=device, dtype=dtype).reshape(t_obs, -1, o)
        if y_dev.shape[1] not in (1, b):
            raise L.PfAmdError(f"observations of shape {tuple(y.shape)} do not broadcast against batch {b}")
        rows = y_dev.shape[1]
        informative = observed if observed is not None else self._observed_flags(y, y_dev)  # host (T_obs,) uint8
        if steps == t_obs:
            y_steps, observed_host = y_dev, informative
        else:
            at = torch.tensor(sched.rows)
            observed_host = torch.zeros(steps, dtype=torch.uint8)
            observed_host[at] = informative
            y_steps = torch.full((steps, rows, o), float("nan"), device=device, dtype=dtype)
            y_steps[at.to(device)] = y_dev

        # recorded states (FilterResult.states; smoothing): the kernels keep a history of `ring` state slots.  The slots
        # become the recorded states themselves (views, no copies), so such a run gets buffers of its own
        keep_states = FilterResult.states_kept(self.record_states)  # None = all
        # the moves whose states / moment rows are reported: those that consumed an observation - and the unobserved
        # sub-steps too when intermediary states are recorded (filters/base.py:207-208); q = move index + 1
        reported = list(range(1, steps + 1)) if (steps == t_obs or self._record_intermediary) else [r + 1 for r in sched.rows]
        if keep_states == 1:
            ring, wanted = 0, reported[-1:]
        else:
            wanted = reported if keep_states is None else reported[-keep_states:]
            ring = max(3, steps - wanted[0] + 1)  # slots for the states wanted[0] .. steps
        taped = ctx.z_tape is not None or ctx.u_tape is not None
        use_graph = ((not taped) and not ring and replay is None and not getattr(self, "_time_kernels", False)
                     and HINTS.graph and not kind.is_user and not self._move_by_move)
        key = (n, b, d, o, steps, rows, dtype, device, self._FILTER_KIND, self._proposal._KERNEL_PROPOSAL,
               self._resampler_kind(), self._seed, float(self._resample_threshold), observed_host.numpy().tobytes(), HINTS.key())
        # a user-defined affine process that opted in (``graph_callable = True`` on the process): the run's whole launch sequence -
        # per move the callable's own torch launches and the library's kernel - is captured ONCE as a hipGraph (torch.cuda.graph)
        # and replayed: the callable's launches are a few microseconds of kernel each, issued eagerly they set the pace of a move
        user_graph = (kind.is_user and bool(getattr(self._model.hidden, "graph_callable", False)) and (not taped) and not ring
                      and replay is None and not getattr(self, "_time_kernels", False) and HINTS.graph
                      and not isinstance(self._move_by_move, (list, tuple)))
        plan = self._fused_plans.get(key) if (use_graph or user_graph) else None
        if plan is None:
            plan = _FusedPlan(self, kind, n, b, d, o, steps, rows, dtype, device, observed_host, ring=ring)
            if use_graph or user_graph:
                if len(self._fused_plans) >= 4:  # a handful of (shape, schedule) combinations at most
                    self._fused_plans.pop(next(iter(self._fused_plans))).destroy()
                self._fused_plans[key] = plan

        # ---- load the inputs into the plan's (persistent) buffers -------------------------------------------------------
        def load_state():
            plan.x[0].copy_(ops.to_soa(x0, self._batched, self._has_event))
            plan.logw[0].copy_(ops.to_cols(state.weights))
            (plan.anc_hist[0] if ring else plan.anc).copy_(state.ancestors32().reshape(b, n))
            plan.ll_total.zero_()

        plan.params.copy_(ctx.params)
        load_state()
        plan.y.copy_(y_steps)
        # fresh Philox draws for every call: the base seed is baked into the (captured) launch arguments, the kernels add
        # the device word `epoch` to it - set here so that base + epoch = this run's draw seed (mod 2^64)
        seed_eff = self._next_draw_seed() if replay is None else replay[0]
        word = (seed_eff - self._seed) & _M64
        plan.epoch.fill_(word - (1 << 64) if word >= (1 << 63) else word)
        a = plan.args
        HINTS.fill(a)  # (a cached plan was keyed by them; a fresh one takes the current ones)
        z_tape = u_tape = None
        if ctx.z_tape is not None:
            z_tape = ctx.z_tape[t_start:t_start + steps].contiguous()
            assert z_tape.shape[0] == steps, "z tape shorter than the number of steps"
        if ctx.u_tape is not None:
            u_tape = ctx.u_tape[t_start:t_start + steps].contiguous()
        elif replay is not None and replay[1] is not None:
            u_tape = replay[1][:steps].contiguous()  # the offsets of the run being repeated
        elif self._resampler_kind() == L.RESAMPLE_SYSTEMATIC:
            # one uniform per (step, filter): T x B numbers drawn up front (a device generator seeded by the run's seed),
            # so no kernel spends a Philox chain on a per-column scalar
            plan.u_gen.manual_seed(seed_eff & 0x7FFFFFFFFFFFFFFF)
            plan.u.uniform_(generator=plan.u_gen)
            u_tape = plan.u
        a.z_tape, a.u_tape = L.ptr(z_tape), L.ptr(u_tape)

        # NB the kernels index tapes / observations by the *local* step 0..steps-1 and draw Philox numbers by it too
        lib = L.load()
        if kind.is_user or self._move_by_move:
            # A user-defined affine process (PF_HID_USER_AFFINE): the caller's mean_scale callable runs ONCE per move, with
            # torch ops on the current particles (a view of the plan's state buffer), into the plan's (loc, scale) planes; the
            # move itself is one run of the fused kernels on the same buffers - no per-move state objects, allocations or
            # copies (the driver's loop over filter() spends ~2x the time of the kernels on those).
            hidden, full = self._model.hidden, x0.shape
            es_u = hidden.event_shape
            # Between the moves of a per-step-route run nothing needs flushing (the next move's launch keeps the books of
            # the state it reads), and a SISR move leaves the partials / scans its successor starts from: intermediate
            # moves run without the bookkeeping launch and - SISR - the successor without the reduce launch
            # (pf_run_hints.resume).  Filters small enough for the column kernel keep self-contained moves (finalize = 1
            # is what makes a call eligible for that one-launch route).
            chained = kind.is_user and (HINTS.route == 1 or n > (HINTS.column_max_n or 2048))
            keep = []
            # An APF move with the optimal proposal and ONE transition scale per filter can prepare its successor's first-stage
            # weights itself, like every step of a built-in model's run (pf_run_hints.prepare_next: they read the new particle
            # and that scale, not the callable's mean) - the successor then starts without the reduce launch as well
            apf_lgo = self._FILTER_KIND == L.FILTER_APF and self._proposal._KERNEL_PROPOSAL == L.PROP_LGO
            prepared_with = None  # the per-column scale the previous move prepared this move's first-stage weights with
            if not kind.is_user and isinstance(self._move_by_move, (list, tuple)):
                # (testing knob, general form: the run as explicit pieces ``(n_steps, finalize)`` on one argument block - a
                # piece without ``finalize`` takes the per-step kernels, a self-contained one of a small filter the column
                # kernel: the workspace's per-filter bookkeeping is what carries a run across the two)
                t_piece = 0
                for n_piece, fin in self._move_by_move:
                    n_piece = steps - t_piece if n_piece is None else n_piece
                    L.check(lib.pf_filter_run(C.byref(a), t_piece, n_piece, fin, L.stream_ptr()), "pf_filter_run")
                    t_piece += n_piece
                assert t_piece == steps, "the pieces do not cover the run"
            def issue_moves():
                nonlocal keep, prepared_with
                for s_ in range(steps if (kind.is_user or not isinstance(self._move_by_move, (list, tuple))) else 0):
                    if not kind.is_user:  # (``_move_by_move``: a built-in model issued the same way - the pieces of one run)
                        L.check(lib.pf_filter_run(C.byref(a), s_, 1, 1, L.stream_ptr()), "pf_filter_run")
                        continue
                    ts = TimeseriesState(t_start + s_, ops.from_soa(plan.x[s_ & 1], self._batched, self._has_event), es_u)
                    loc, scale, a.user_dt = self._user_mean_scale(hidden, ts)
                    v = ops.to_soa(loc.to(dtype).expand(full), self._batched, self._has_event)
                    # (already a (D, B, N) plane - an unbatched scalar state's loc: read in place, kept alive past the launch)
                    loc_p = v if v.is_contiguous() else plan.user_loc.copy_(v)
                    percol = self._scale_per_column(scale, full, dtype)
                    if percol is not None:  # a state-independent diffusion: one scale per filter and component, no plane to fill
                        scale_p = percol
                    else:
                        v = ops.to_soa(scale.to(dtype).expand(full), self._batched, self._has_event)
                        scale_p = v if v.is_contiguous() else plan.user_scale.copy_(v)
                    keep = [loc_p, scale_p]
                    a.user_loc, a.user_scale = loc_p.data_ptr(), scale_p.data_ptr()
                    a.user_scale_per_column = 0 if percol is None else 1
                    last_move = s_ == steps - 1
                    # (prepared first-stage weights are only taken when this move's scale IS the one they were computed with - a
                    # time-dependent diffusion hands over another tensor, and the move re-reduces like any other)
                    a.hints.resume = 1 if (chained and s_ > 0 and (self._FILTER_KIND == L.FILTER_SISR or
                                                                   (prepared_with is not None and percol is prepared_with))) else 0
                    prepare = bool(chained and apf_lgo and percol is not None and not last_move and observed_host[s_ + 1])
                    a.hints.prepare_next = 1 if prepare else 0
                    prepared_with = percol if prepare else None
                    L.check(lib.pf_filter_run(C.byref(a), s_, 1, 1 if (last_move or not chained) else 0, L.stream_ptr()), "pf_filter_run")

            captured = False
            if user_graph and plan.runs >= 1 and not plan.user_graph_failed:
                # (the first run of a configuration is issued eagerly: it warms up whatever the callable initialises lazily, and a
                # configuration used once never pays for a capture).  A graph is tied to the tensors the callable read when it was
                # captured: another parameter tensor (not an in-place update of the same one) captures again
                sig = (id(ctx),) + tuple(p.data_ptr() for p in hidden.parameters if isinstance(p, torch.Tensor))
                if plan.user_graph is None or plan.user_graph_sig != sig:
                    graph = torch.cuda.CUDAGraph()
                    # (the per-filter scale rows derived from the callable's scale are remembered across moves by the tensor's
                    # identity and version: inside a capture they must be COMPUTED - once, by the first move - so that a replay
                    # derives them from the parameter's current values; the entry is dropped again afterwards)
                    self._percol_cache = None
                    try:
                        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                            issue_moves()
                        plan.user_graph, plan.user_graph_sig, plan.user_graph_keep = graph, sig, (list(keep), self._percol_cache)
                        self._percol_cache = None
                    except Exception as e:  # the callable does something a capture cannot record (a host read, an allocation-
                        # dependent branch): this configuration stays eager
                        import warnings

                        plan.user_graph, plan.user_graph_failed = None, True
                        self._percol_cache = None
                        warnings.warn(f"graph_callable: capturing the callable failed ({type(e).__name__}: {e}); the run is issued eagerly")
                        torch.cuda.synchronize()
                if plan.user_graph is not None:
                    plan.user_graph.replay()
                    captured = True
            if not captured:
                issue_moves()
            if kind.is_user:
                a.user_loc, a.user_scale = plan.user_loc.data_ptr(), plan.user_scale.data_ptr()
                a.user_scale_per_column, a.hints.resume, a.hints.prepare_next = 0, 0, 0
                self._keep_planes = keep
        elif getattr(self, "_time_kernels", False):
            kms = (C.c_float * 3)()
            L.check(lib.pf_filter_run_timed(C.byref(a), 0, steps, 1, L.stream_ptr(), kms), "pf_filter_run_timed")
            self.kernel_ms = tuple(kms)
        elif use_graph and (plan.graph is not None or plan.runs >= 1):
            # (a configuration's first run launches directly: capturing + instantiating T kernel nodes costs more host time
            # than issuing them once - an SMC^2 rejuvenation re-filters t observations exactly once per t)
            if plan.graph is None:
                h = C.c_void_p(None)
                L.check(lib.pf_filter_graph_create(C.byref(a), 0, steps, 1, L.stream_ptr(), C.byref(h)), "pf_filter_graph_create")
                plan.graph = h
            L.check(lib.pf_filter_graph_launch(plan.graph, L.stream_ptr()), "pf_filter_graph_launch")
        else:
            L.check(lib.pf_filter_run(C.byref(a), 0, steps, 1, L.stream_ptr()), "pf_filter_run")
        plan.runs += 1
        # (this driver carries cluster-route runs only in its taped / timed / piecewise modes - tests and tools; plain runs of
        # that size take the lean driver: a launch that gave up is re-issued on the per-step route there and here alike)
        if (not ring and not kind.is_user and HINTS.cluster_takes(n, b, self._resampler_kind() == L.RESAMPLE_SYSTEMATIC)
                and int(plan.status.item()) != 0):
            self._cluster_gave_up(plan)
            load_state()
            a.hints.route = 1  # PF_ROUTE_PER_STEP
            L.check(lib.pf_filter_run(C.byref(a), 0, steps, 1, L.stream_ptr()), "pf_filter_run")
        self._last_run = dict(plan=plan, z=z_tape, u=u_tape, ws=plan.ws, seed_eff=seed_eff)  # keep device buffers alive

        # ---- hand the results over in the reference's shapes (copies: a cached plan's buffers are reused) ------------
        shape_md = (lambda t: t if self._batched else t[:, 0])
        means_v, vars_v = shape_md(plan.means), shape_md(plan.vars)  # (steps + 1, [B], D): copied by the moment log
        ll_steps, ll_total = plan.ll_steps.clone(), plan.ll_total.clone()
        self._last_run["ll_steps"] = ll_steps
        es = self._model.hidden.event_shape

        def state_of(q, x_soa, lw, anc32):
            """The state after move q - 1 (q >= 1) as the reference's object."""
            ll_q = ll_steps[q - 1] if self._batched else ll_steps[q - 1, 0]
            st = ParticleFilterCorrection(
                TimeseriesState(t_start + q, ops.from_soa(x_soa, self._batched, self._has_event), es),
                ops.from_cols(lw, self._batched), ll_q, None,
                _moments=(means_v[q].clone(), vars_v[q].clone()), _anc32=(anc32, self._batched),
            )
            return st

        if ring:
            recorded = [state_of(q, plan.x_hist[q % ring], plan.logw_hist[q % ring], plan.anc_hist[q % ring]) for q in wanted]
            last = recorded[-1]
        else:
            slot = steps % 2
            last = state_of(steps, plan.x[slot].clone(), plan.logw[slot].clone(), plan.anc.clone())
            recorded = None
        if len(reported) == steps:
            sel_m, sel_v = means_v, vars_v
        else:  # the initial state's row, then the row of every reported move
            keep = torch.tensor([0] + reported, device=device)
            sel_m, sel_v = means_v[keep], vars_v[keep]
        result._extend_fused(sel_m, sel_v, ll_total if self._batched else ll_total[0], last, states=recorded)
        self._last_run["rows"] = (sel_m, sel_v)  # every reported row of this run, whatever window the result's log keeps
        return result

    # ------------------------------------------------------------------------------------------------------------
    # smoothing over recorded states (particle/base.py:105-157)
    # ------------------------------------------------------------------------------------------------------------
    def _history(self, states):
        """``(x_hist (S, D, B, N), logw_hist (S, B, N), anc_hist (S, B, N) int32)`` of consecutive recorded states - views
        of the kernels' history ring when the states came from a fused run, else stacked copies."""
        xs = [ops.to_soa(s.timeseries_state.value, self._batched, self._has_event) for s in states]
        ws = [ops.to_cols(s.weights) for s in states]
        an = []
        for s in states:
            an.append(s.ancestors32())  # (the kernels' own buffer - a slot of their history ring - while it is current)

        def stacked(ts):
            step = ts[0].numel() * ts[0].element_size()
            if all(t.is_contiguous() for t in ts) and all(ts[k].data_ptr() == ts[0].data_ptr() + k * step for k in range(len(ts))):
                return torch.as_strided(ts[0], (len(ts),) + tuple(ts[0].shape), (ts[0].numel(),) + tuple(ts[0].stride()))
            return torch.stack(ts, 0).contiguous()

        return stacked(xs), stacked(ws), stacked(an)

    def set_smoothing_tape(self, u: Optional[torch.Tensor]):
        """Parity mode for ``smooth(..., "ffbs")``: the uniforms of the backward draws, ``(S - 1, N, [B])`` (row t serves
        the draw of state t given state t + 1) - ``None`` restores Philox."""
        self._ffbs_u = u

    def smooth(self, states, method: str = "ffbs") -> torch.Tensor:
        """Smoothed trajectories ``(S, N, [B], [D])`` from recorded states: ``"fl"`` - every particle of the last state
        traced back along its ancestors (:136-152); ``"ffbs"`` - forward filtering, backward simulation (:105-134).  Both are
        one kernel launch over the state history for built-in models on the GPU; ``"ffbs"`` with user callables evaluates
        the reference's (N, N) logits with torch ops."""
        states = list(states)
        low = method.lower()
        if low not in ("ffbs", "fl"):
            raise NotImplementedError(f"Currently do not support '{method}'!")
        x_last = states[-1].timeseries_state.value
        on_gpu = x_last.is_cuda
        if low == "fl":
            if not on_gpu:
                raise L.PfAmdError("pyfilter_amd runs on MI355X only: smoothing needs states on the GPU")
            x_hist, _, anc_hist = self._history(states)
            out = ops.smooth_fixed_lag(x_hist, anc_hist)
            return torch.stack([ops.from_soa(o, self._batched, self._has_event) for o in out.unbind(0)], 0)
        # ffbs: the last state resampled by the filter's resampler, then the backward draws
        w_last = states[-1].weights
        idx = self._resampler(w_last)
        from ..utils import batched_gather

        start = batched_gather(x_last, idx, 0)
        ctx = self._ensure_context()
        if ctx is not None and on_gpu and not ctx.kind.is_user:
            x_hist, w_hist, _ = self._history(states)
            u = getattr(self, "_ffbs_u", None)
            if u is not None:
                s1 = len(states) - 1
                u = u.to(device=x_hist.device, dtype=x_hist.dtype).reshape(s1, x_hist.shape[3], x_hist.shape[2]).permute(0, 2, 1).contiguous()
            out = ops.smooth_ffbs(ctx.kind, ctx.params, x_hist, w_hist, ops.to_soa(start, self._batched, self._has_event),
                                  u, self._next_draw_seed())
            return torch.stack([ops.from_soa(o, self._batched, self._has_event) for o in out.unbind(0)], 0)
        return self._ffbs_with_callables(states, start)

    def _ffbs_with_callables(self, states, start):
        """User-defined models: the reference's own evaluation (an (N, N, [B]) logits tensor per step) with torch ops."""
        from torch.distributions import Categorical

        res = [start]
        has_event = self._has_event
        for state in reversed(states[:-1]):
            density = self._model.hidden.build_density(state.timeseries_state)
            w_state = density.log_prob(res[-1].unsqueeze(1))
            weights = state.weights.unsqueeze(0) + w_state
            if self._batched:
                weights = weights.moveaxis(1, 2)
            indices = Categorical(logits=weights).sample()
            if has_event:
                indices = indices.unsqueeze(-1).expand(self.particles + self._model.hidden.event_shape)
            res.append(state.timeseries_state.value.gather(0, indices))
        return torch.stack(res[::-1], dim=0)


class _OnlineRun:
    """The filters of an observation-by-observation loop (``SMC2.step``, ``smc2.py:53-65``) as ONE run issued piece by piece on one
    argument block (include/pf_amd.h: ``pf_filter_run(args, m, 1, 1)``, m = 0, 1, ...): two state slots the kernels alternate
    between, the moves' moment rows and increments written straight into arrays of ``ROWS`` moves - per observation the host sets
    the observation's address and a fresh draw seed, makes one ``pf_filter_run`` call and one ``pf_theta_step`` call (weights,
    ESS, running log-likelihood, host slot) and polls the slot.  Nothing else is built per observation: the ``FilterResult`` gets
    its rows and its latest state when somebody looks (``flush``: a rejuvenation, a full array, the caller's own access).

    APF on a built-in model, filters on the batch dimension, one shared observation row, no recorded states / tapes."""

    ROWS = 64

    def __init__(self, filt, result):
        ctx = filt._ensure_context()
        kind = ctx.kind
        st = result.latest_state
        x = ops.to_soa(st.timeseries_state.value, True, filt._has_event)
        d, b, n = x.shape
        self.filt, self.result, self.kind = filt, result, kind
        self.d, self.b, self.n, self.o = d, b, n, kind.obs_dim
        self.dtype, self.device = x.dtype, x.device
        self.plan = plan = _SingleStepPlan(filt, kind, n, b, d, kind.obs_dim, 1, x.dtype, x.device)
        k = self.ROWS
        self.xl = [torch.empty((d + 1, b, n), device=x.device, dtype=x.dtype) for _ in range(2)]
        self.anc = torch.empty((b, n), device=x.device, dtype=torch.int32)
        self.rows = torch.empty((2, k + 1, b, d), device=x.device, dtype=x.dtype)
        self.ll = torch.zeros((k, b), device=x.device, dtype=x.dtype)
        self.stats = torch.empty((k, 2), device=x.device, dtype=x.dtype)
        self.scratch_total = torch.zeros(b, device=x.device, dtype=x.dtype)
        self.es = plan.elem_size
        self.m, self.t, self.synced = 0, 0, None
        a = plan.args
        a.y_rows, a.observed, a.observed_dev, a.step_counter = 1, None, None, None
        a.anc = self.anc.data_ptr()
        a.means, a.vars = self.rows[0].data_ptr(), self.rows[1].data_ptr()
        a.ll_steps, a.ll_total = self.ll.data_ptr(), self.scratch_total.data_ptr()
        a.z_tape, a.u_tape = None, None
        self._point_slots()
        self._lib = L.load()
        self._theta_step, self._observe = self._lib.pf_theta_step, self._lib.pf_filter_observe
        self._code = L.dtype_code(x.dtype)
        self._ll_ptr, self._stats_ptr = self.ll.data_ptr(), self.stats.data_ptr()
        self._status_ptr = plan.status.data_ptr()
        self.hints_key, self._stream = None, None

    @staticmethod
    def applies(filt, result) -> bool:
        from .apf import APF

        if not isinstance(filt, APF) or not filt._batched or not HINTS.fused_step or len(result._states) == 0:
            return False
        if type(filt).filter is not ParticleFilter.filter:  # (a subclass that wraps filter() - e.g. to inject draws - must be called)
            return False
        x = result.latest_state.timeseries_state.value
        return (filt._fused_capable(x.device) and int(filt._model.observe_every_step) == 1 and not filt._record_intermediary
                and not filt._kernel_kind().is_user and FilterResult.states_kept(filt.record_states) == 1
                and filt._resampler_kind() == L.RESAMPLE_SYSTEMATIC)

    def _point_slots(self):
        a, d = self.plan.args, self.d
        a.x[0], a.x[1] = self.xl[0].data_ptr(), self.xl[1].data_ptr()
        a.logw[0], a.logw[1] = self.xl[0][d].data_ptr(), self.xl[1][d].data_ptr()

    def _in_sync(self) -> bool:
        """The result's latest state is still the one this run flushed (nobody replaced or edited it since)."""
        s = self.synced
        if s is None or len(self.result._states) == 0 or self.result._states[-1] is not s[0]:
            return False
        st = s[0]
        x, w = st.timeseries_state.value, st["_w"]
        return x is s[1] and w is s[2] and x._version == s[3] and w._version == s[4]

    def _attach(self):
        """Slot 0 <- the result's latest state (three copies: after a rejuvenation, or at the loop's start)."""
        st = self.result.latest_state
        d = self.d
        xl = getattr(st, "_xl", None)
        if xl is not None and xl.shape == self.xl[0].shape and xl.dtype == self.dtype:
            self.xl[0].copy_(xl)
        else:
            self.xl[0][:d].copy_(ops.to_soa(st.timeseries_state.value, True, self.filt._has_event))
            self.xl[0][d].copy_(ops.to_cols(st.weights))
        self.m, self.t = 0, int(st.timeseries_state.time_index)
        x_, w_ = st.timeseries_state.value, st["_w"]
        self.synced = (st, x_, w_, x_._version, w_._version)

    def observe(self, y: torch.Tensor, w: torch.Tensor, slot):
        """One observation: the filters' move, ``w += ll`` and its (ESS, all finite) pair on the host.  Returns the pair; the
        device-side statistics row of the move just made is ``self.last_stats`` (array, row)."""
        filt, plan = self.filt, self.plan
        a = plan.args
        ctx = filt._ensure_context()
        stream = L.stream_ptr()
        if stream != self._stream:  # (pieces m > 0 rely on stream order behind piece m - 1: another stream starts another run)
            self.flush()
            self._stream = stream
        if self.m == 0 and not self._in_sync():
            self._attach()
        if y.dtype != self.dtype or not y.is_cuda or not y.is_contiguous():
            y = y.to(device=self.device, dtype=self.dtype).contiguous()
        hk = HINTS.key()
        if self.hints_key != hk:
            HINTS.fill(a)
            self.hints_key = hk
        m, es, b = self.m, self.es, self.b
        a.model.params = ctx.params.data_ptr()
        a.y = y.data_ptr() - m * self.o * es
        a.seed = filt._next_draw_seed()
        if ctx.z_tape is not None or ctx.u_tape is not None:  # (parity mode: row `t` of the tapes is piece m's)
            zt = None if ctx.z_tape is None else ctx.z_tape[self.t]
            ut = None if ctx.u_tape is None else ctx.u_tape[self.t]
            a.z_tape = None if zt is None else zt.data_ptr() - m * self.d * b * self.n * es
            a.u_tape = None if ut is None else ut.data_ptr() - m * b * es
        plan.generation = a.hints.cluster_generation = plan.generation % 0xFFFFF + 1
        total = self.result._loglikelihood
        seq = slot.seq + 1
        # the move and the theta update in one call (pf_filter_observe = pf_filter_run + pf_theta_step under the move's status word)
        L.check(self._observe(plan.args_ref, m, 1, 1, w.data_ptr(), self._ll_ptr + m * b * es, self._stats_ptr + 2 * m * es, slot.ptr, seq,
                              total.data_ptr(), stream), "pf_filter_observe")
        slot.seq = seq
        pair = slot.wait()
        if slot.status:  # a column-cluster launch gave up (nothing was updated): the piece again on the per-step route, same draws
            filt._cluster_gave_up(plan)
            self.scratch_total.zero_()
            a.hints.route = 1
            L.check(plan.run(plan.args_ref, m, 1, 1, stream), "pf_filter_run")
            self.hints_key = None
            seq = slot.seq + 1
            L.check(self._theta_step(w.data_ptr(), self._ll_ptr + m * b * es, b, self._code, self._stats_ptr + 2 * m * es, slot.ptr, seq,
                                     total.data_ptr(), None, stream), "pf_theta_step")
            slot.seq = seq
            pair = slot.wait()
        self._keep = (y, ctx.params)
        self.last_stats = (self.stats, m)  # (the move's statistics row: a flush replaces self.stats)
        self.m, self.t = m + 1, self.t + 1
        if self.m == self.ROWS:
            self.flush()
        return pair

    def flush(self):
        """The moves made since the last flush join the ``FilterResult``: their moment rows and - as an object of its own, with its
        own buffers - the latest state."""
        m = self.m
        if m == 0:
            return
        res, filt, d = self.result, self.filt, self.d
        res._moments.extend(self.rows[0][1:m + 1], self.rows[1][1:m + 1])
        slot = m & 1
        xl = self.xl[slot].clone()
        anc = self.anc.clone()
        last = ParticleFilterCorrection(
            TimeseriesState(self.t, ops.from_soa(xl[:d], True, filt._has_event), filt._model.hidden.event_shape),
            ops.from_cols(xl[d], True), self.ll[m - 1].clone(), None,
            _moments=(self.rows[0][m].clone(), self.rows[1][m].clone()), _anc32=(anc, True))
        last._xl = xl
        res._states.append(last)
        if slot:  # the run goes on from slot 0
            self.xl.reverse()
            self._point_slots()
        # (the caller keeps VIEWS of the statistics rows - SMC2State.ess -: a used array is replaced, never rewritten)
        self.stats = torch.empty_like(self.stats)
        self._stats_ptr = self.stats.data_ptr()
        self.m = 0
        x_, w_ = last.timeseries_state.value, last["_w"]
        self.synced = (last, x_, w_, x_._version, w_._version)


class _BlockResult:
    """What ``filter_block`` hands back on its lean route: the run's rows (row 0 = the incoming state), its total
    log-likelihood and final state behind ``FilterResult``'s read interface - no moment log, no state deque."""

    def __init__(self, means, variances, loglikelihood, last_state):
        self.filter_means, self.filter_variance = means, variances
        self.loglikelihood, self.latest_state = loglikelihood, last_state
        self.block_rows = (means[1:], variances[1:])  # the moves' own rows

    @property
    def states(self):
        return [self.latest_state]


class _SingleStepPlan:
    """Scratch + launch arguments of the fused *single-step* move behind ``filter()`` (the online / SMC^2 entry point):
    the kernels read the incoming state's own buffers and write freshly allocated ones that become the new state -
    no staging copies, four launches (observed flag, reduce, step, bookkeeping) instead of the ~10 of the step-by-step route."""

    def __init__(self, filt, kind, n, b, d, o, rows, dtype, device):
        self.cdf = torch.empty((b, n), device=device, dtype=dtype)
        self.pos = torch.empty((b, n), device=device, dtype=dtype)
        self.ws = L.new_workspace(n, b, device)
        self.status = torch.zeros(1, device=device, dtype=torch.int32)  # pf_filter_args.status: sticky, cleared by _cluster_gave_up
        self.rows = rows
        self.xl = None  # (``_filter_block_lean``: the second state slot of a multi-move run, allocated on first use)
        self.u_gen = None
        a = L.PfFilterArgs()
        a.model = ops.make_model_struct(kind, filt._ctx.params)
        a.filter, a.proposal, a.resampler = filt._FILTER_KIND, filt._proposal._KERNEL_PROPOSAL, filt._resampler_kind()
        a.dtype = L.dtype_code(dtype)
        a.N, a.B = n, b
        a.ess_threshold = float(filt._resample_threshold) / float(n)
        a.seed = filt._seed
        a.cdf, a.pos = self.cdf.data_ptr(), self.pos.data_ptr()
        a.y, a.y_rows, a.observed, a.observed_dev = None, rows, None, None  # (flags: derived from y by the run)
        a.step_counter = None
        a.ws, a.ws_bytes = self.ws.data_ptr(), self.ws.numel()
        a.status = self.status.data_ptr()
        self.args = a
        self.args_ref = C.byref(a)
        self.run = L.load().pf_filter_run
        self.n, self.b, self.d, self.o, self.rows, self.dtype, self.device = n, b, d, o, rows, dtype, device
        self.kind, self.rs_kind, self.thr, self.hints_key = kind, None, None, None
        self.elem_size = torch.empty((), dtype=dtype).element_size()
        self._pool = None
        self._pool_next = 0
        self._view_geo = None
        self.generation = 0  # pf_run_hints.cluster_generation of the latest cluster launch on self.ws (zero-filled: new_workspace)
        self.chain = None  # the resume token: (piece, x_out, lw_out, their versions, hints) of the latest per-step-route online move

    _STATS_POOL = 64

    def zeroed_stats(self, batched: bool):
        """One move's zeroed statistics block - means (2, B, D) | variances (2, B, D) | ll (B) | total (B), ``4 D + 2`` rows of
        ``B`` - cut from a pool that is zeroed 64 moves at a time (one fill launch per 64 moves instead of one per move).
        Returns the NEW state's views (mean, variance, log-likelihood increment - in the state's own shapes) and the block's
        device address; the kernels get pointers computed from it.  A block is handed out ONCE: the states keep views of it, and an
        exhausted pool is replaced, never refilled."""
        if self._pool is None or self._pool_next == self._STATS_POOL:
            d, b = self.d, self.b
            pool = torch.zeros((self._STATS_POOL, 4 * d + 2, b), device=self.device, dtype=self.dtype)
            mean = pool[:, d:2 * d].reshape(self._STATS_POOL, b, d)       # (the block's means[1], variances[1], ll rows)
            var = pool[:, 3 * d:4 * d].reshape(self._STATS_POOL, b, d)
            ll = pool[:, 4 * d]
            if not batched:
                mean, var, ll = mean[:, 0], var[:, 0], ll[:, 0]
            # (NOT unbound into 3 x 64 views here: that many tracked objects created at once push CPython's generation-0 counter
            # over its threshold again and again, and every hundredth of those collections is a full one - 40 ms in a process
            # that has imported torch; three index calls per move keep the allocation rate flat)
            self._pool = (pool, mean, var, ll, pool.data_ptr(), (4 * d + 2) * b * self.elem_size)
            self._pool_next = 0
        i = self._pool_next
        self._pool_next = i + 1
        _, mean, var, ll, base, stride = self._pool
        return mean[i], var[i], ll[i], base + i * stride

    def state_views(self, x_out: torch.Tensor, lw_out: torch.Tensor, batched: bool, has_event: bool):
        """The reference's ``(N, [B], [D])`` / ``(N, [B])`` views of the kernels' ``(D, B, N)`` / ``(B, N)`` buffers - one
        ``as_strided`` each (``ops.from_soa / from_cols`` spell them as a permute and up to two selects)."""
        geo = self._view_geo
        if geo is None or geo[0] != (batched, has_event):
            xv, wv = ops.from_soa(x_out, batched, has_event), ops.from_cols(lw_out, batched)
            geo = self._view_geo = ((batched, has_event), tuple(xv.shape), tuple(xv.stride()), tuple(wv.shape), tuple(wv.stride()))
            return xv, wv
        return x_out.as_strided(geo[1], geo[2]), lw_out.as_strided(geo[3], geo[4])


class _FusedPlan:
    """Persistent device buffers + launch arguments (+ the captured hipGraph) of one fused-run configuration.  Keeping
    them across ``batch_filter`` calls means repeated runs (PMMH / SMC^2 re-filter the same data many times,
    ``inference/batch/mcmc/utils.py:55``) replay one graph instead of issuing 2 T kernel launches from the host."""

    def __init__(self, filt, kind, n, b, d, o, steps, rows, dtype, device, observed_host, ring=0):
        self.ring = ring
        self.runs = 0
        if ring:  # state history (pf_filter_args.ring): slot q % ring holds the state after move q - 1
            self.x_hist = torch.empty((ring, d, b, n), device=device, dtype=dtype)
            self.logw_hist = torch.empty((ring, b, n), device=device, dtype=dtype)
            self.anc_hist = torch.empty((ring, b, n), device=device, dtype=torch.int32)
            self.x, self.logw, self.anc = (self.x_hist[0], self.x_hist[1]), (self.logw_hist[0], self.logw_hist[1]), self.anc_hist
        else:
            self.x = (torch.empty((d, b, n), device=device, dtype=dtype), torch.empty((d, b, n), device=device, dtype=dtype))
            self.logw = (torch.empty((b, n), device=device, dtype=dtype), torch.empty((b, n), device=device, dtype=dtype))
            self.anc = torch.empty((b, n), device=device, dtype=torch.int32)
        self.cdf = torch.empty((b, n), device=device, dtype=dtype)
        self.pos = torch.empty((b, n), device=device, dtype=dtype)
        self.y = torch.empty((steps, rows, o), device=device, dtype=dtype)
        self.means = torch.empty((steps + 1, b, d), device=device, dtype=dtype)
        self.vars = torch.empty_like(self.means)
        self.ll_steps = torch.zeros((steps, b), device=device, dtype=dtype)
        self.ll_total = torch.zeros(b, device=device, dtype=dtype)
        self.epoch = torch.zeros(1, device=device, dtype=torch.int64)
        self.u = torch.empty((steps, b), device=device, dtype=dtype)
        self.u_gen = torch.Generator(device=device)
        self.params = torch.empty_like(filt._ctx.params)
        self.ws = L.new_workspace(n, b, device)
        self.status = torch.zeros(1, device=device, dtype=torch.int32)  # pf_filter_args.status (see _SingleStepPlan)
        self.observed_host = observed_host
        self.user_loc = self.user_scale = None
        if kind.is_user:  # the callable's one-step mean / scale of the current particles, refreshed before every move
            self.user_loc = torch.empty((d, b, n), device=device, dtype=dtype)
            self.user_scale = torch.empty((d, b, n), device=device, dtype=dtype)
        self.graph = None
        self.user_graph = self.user_graph_sig = self.user_graph_keep = None  # torch.cuda.CUDAGraph of a user-affine run (graph_callable)
        self.user_graph_failed = False

        a = L.PfFilterArgs()
        a.model = ops.make_model_struct(kind, self.params)
        a.filter, a.proposal, a.resampler = filt._FILTER_KIND, filt._proposal._KERNEL_PROPOSAL, filt._resampler_kind()
        a.dtype = L.dtype_code(dtype)
        a.N, a.B = n, b
        a.ess_threshold = float(filt._resample_threshold) / float(n)
        a.seed = filt._seed
        a.x[0], a.x[1] = self.x[0].data_ptr(), self.x[1].data_ptr()
        a.logw[0], a.logw[1] = self.logw[0].data_ptr(), self.logw[1].data_ptr()
        a.anc, a.cdf, a.pos = self.anc.data_ptr(), self.cdf.data_ptr(), L.ptr(self.pos)
        a.y, a.y_rows, a.observed = self.y.data_ptr(), rows, observed_host.data_ptr()
        a.z_tape, a.u_tape = None, None
        a.means, a.vars = self.means.data_ptr(), self.vars.data_ptr()
        a.ll_steps, a.ll_total = self.ll_steps.data_ptr(), self.ll_total.data_ptr()
        a.step_counter = self.epoch.data_ptr()
        a.ws, a.ws_bytes = self.ws.data_ptr(), self.ws.numel()
        a.status = self.status.data_ptr()
        a.ring = ring
        a.user_loc, a.user_scale = L.ptr(self.user_loc), L.ptr(self.user_scale)
        self.args = a

    def destroy(self):
        if self.graph is not None:
            torch.cuda.synchronize()
            L.load().pf_filter_graph_destroy(self.graph)
            self.graph = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
