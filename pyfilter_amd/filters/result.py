"""``FilterResult``: what a filter run leaves behind - the time series of filter means / variances (incl. the initial
state), the running log-likelihood and the recorded states.

Interface and wire format are the reference's (``pyfilter/filters/result.py:14-164``: ``filter_means``,
``filter_variance``, ``loglikelihood``, ``states``, ``latest_state``, ``append``, ``exchange``, ``resample``,
``state_dict`` / ``load_state_dict`` with the ``tensor_deque_<maxlen>__filter_means`` keys of
``pyfilter/container.py:113-139``).  The storage is this library's own: the reference keeps one small tensor per time
step in Python deques and loops over them for every batch-dim move; here both moment series live in ONE device buffer
laid out ``(filters, time, 2 * dim)`` - a filter's whole history is one contiguous column - so

* the fused kernels' ``(T + 1, B, D)`` rows are adopted with one strided copy,
* ``resample`` (gather whole filters) and ``exchange`` (masked overwrite) of the entire history are ONE
  ``pf_columns_gather`` / ``pf_columns_exchange`` launch each (``include/pf_amd.h``),
* a bounded history (``record_moments=<int>`` / ``False``) is a sliding window over the same buffer.
"""
from collections import OrderedDict
from copy import deepcopy
from typing import Generic, List, Optional, TypeVar

import torch

from ..container import BoolOrInt, TensorContainer, make_dequeue
from .state import Correction

TCorrection = TypeVar("TCorrection", bound=Correction)


def _deque_maxlen(spec: BoolOrInt) -> Optional[int]:
    """``record_moments`` -> the reference's deque length (``container.py:10-18``): ``False`` 1, ``True`` / ``None``
    unbounded, an int that many."""
    return make_dequeue(spec).maxlen


class MomentLog:
    """Time series of (mean, variance) rows of B parallel filters in one ``(B, capacity, 2 * dim)`` buffer.

    A row is ``(mean[dim], variance[dim])``; ``rows`` of them are live, ending at ``_stop``.  With a ``maxlen`` the live
    window slides; the buffer is compacted (one copy of ``maxlen - 1`` rows) whenever the window reaches its end, so an
    online run of any length costs O(maxlen) memory and amortised O(1) copies per step."""

    # Rows appended one at a time (``append``: the online ``filter()`` loop) are only REMEMBERED - the state's own (mean,
    # variance) tensors - and written into the buffer ``_PENDING_MAX`` at a time (two stacks + one strided copy each) or when
    # somebody looks: an online move then costs no launch for its moment row (it cost two strided copies).
    _PENDING_MAX = 64

    def __init__(self, maxlen: Optional[int]):
        self.maxlen = maxlen
        self._buf_: Optional[torch.Tensor] = None
        self._stop_ = 0
        self._rows_ = 0
        self._pending = []
        self._row_shape: Optional[torch.Size] = None  # shape of ONE state's mean, e.g. (B, D), (D,), (B,), ()
        self._batched = False

    # (every reader / writer of the buffer's bookkeeping sees the remembered rows written first)
    def _flush(self):
        if self._pending:
            rows, self._pending = self._pending, []
            self.extend(torch.stack([m for m, _ in rows]), torch.stack([v for _, v in rows]))

    @property
    def _buf(self):
        self._flush()
        return self._buf_

    @_buf.setter
    def _buf(self, value):
        self._buf_ = value

    @property
    def _stop(self):
        self._flush()
        return self._stop_

    @_stop.setter
    def _stop(self, value):
        self._stop_ = value

    @property
    def rows(self):
        self._flush()
        return self._rows_

    @rows.setter
    def rows(self, value):
        self._rows_ = value

    # ---- geometry -----------------------------------------------------------------------------------------------
    def _adopt_shape(self, mean: torch.Tensor, batched: bool):
        self._row_shape, self._batched = mean.shape, batched
        b = mean.shape[0] if batched else 1
        dim = max(1, mean.numel() // b)
        cap = 64 if self.maxlen is None else max(2, 2 * self.maxlen)
        self._buf = torch.empty((b, cap, 2 * dim), dtype=mean.dtype, device=mean.device)

    @property
    def _dim(self) -> int:
        return self._buf.shape[2] // 2

    def _room_for(self, extra: int):
        """Makes the buffer hold ``extra`` more rows behind ``_stop`` (sliding / growing as needed)."""
        if self.maxlen is not None and extra >= self.maxlen:
            self._stop, self.rows = 0, 0  # everything live is about to fall out of the window
        keep = self.rows if self.maxlen is None else min(self.rows, max(0, self.maxlen - extra))
        cap = self._buf.shape[1]
        if self._stop + extra <= cap:
            return
        need = keep + extra
        src = self._buf[:, self._stop - keep:self._stop]
        new_cap = max(need, 2 * cap) if self.maxlen is None else max(need, 2 * self.maxlen)
        if new_cap > cap:
            grown = torch.empty((self._buf.shape[0], new_cap, self._buf.shape[2]), dtype=self._buf.dtype, device=self._buf.device)
            grown[:, :keep] = src
            self._buf = grown
        elif keep:
            self._buf[:, :keep] = src.clone()
        self._stop, self.rows = keep, keep

    def _canon(self, t: torch.Tensor, lead: int) -> torch.Tensor:
        """``(lead..., *row_shape)`` -> ``(B, lead..., dim)``."""
        b = self._buf.shape[0]
        if lead == 0:
            return t.reshape(b, self._dim)
        steps = t.shape[0]
        t = t.reshape(steps, b, self._dim)
        return t.permute(1, 0, 2)

    # ---- writing ------------------------------------------------------------------------------------------------
    def append(self, mean: torch.Tensor, var: torch.Tensor, batched: bool):
        if self._buf_ is None:
            self._adopt_shape(mean, batched)
        if mean.shape != self._row_shape or var.shape != mean.shape:  # (a row of another shape: written at once, as it always was)
            self._room_for(1)
            d = self._dim
            self._buf[:, self._stop, :d] = self._canon(mean, 0)
            self._buf[:, self._stop, d:] = self._canon(var, 0)
            self._advance(1)
            return
        pending = self._pending
        pending.append((mean, var))
        if self.maxlen is not None and len(pending) > self.maxlen:
            del pending[0]  # (a bounded history: the remembered rows alone fill the window, an older one can never be seen)
        elif len(pending) >= self._PENDING_MAX:
            self._flush()

    def extend(self, means: torch.Tensor, variances: torch.Tensor):
        """Adopts ``steps`` rows at once: ``means`` / ``variances`` are ``(steps, *row_shape)`` (the fused kernels' rows)."""
        steps = means.shape[0]
        if steps == 0:
            return
        if self.maxlen is not None and steps > self.maxlen:
            means, variances = means[-self.maxlen:], variances[-self.maxlen:]
            steps = self.maxlen
        self._room_for(steps)
        d = self._dim
        dst = self._buf[:, self._stop:self._stop + steps]
        dst[..., :d] = self._canon(means, 1)
        dst[..., d:] = self._canon(variances, 1)
        self._advance(steps)

    def _advance(self, k: int):
        self._stop += k
        self.rows = self.rows + k if self.maxlen is None else min(self.maxlen, self.rows + k)

    # ---- reading ------------------------------------------------------------------------------------------------
    def _series(self, lo: int, hi: int) -> torch.Tensor:
        if self._buf is None or self.rows == 0:
            return torch.tensor([])
        live = self._buf[:, self._stop - self.rows:self._stop, lo:hi]  # (B, rows, dim)
        return live.permute(1, 0, 2).reshape((self.rows,) + tuple(self._row_shape))

    def means(self) -> torch.Tensor:
        return self._series(0, self._dim if self._buf is not None else 0)

    def variances(self) -> torch.Tensor:
        return self._series(self._dim, 2 * self._dim) if self._buf is not None else torch.tensor([])

    # ---- whole-filter moves along the batch dim: one kernel launch for the entire history -----------------------
    def _live_columns(self) -> torch.Tensor:
        """The live window as a contiguous ``(1, B, rows * 2 dim)`` array of columns (a view when the window starts the
        buffer and fills it, else a compacted copy that becomes the buffer)."""
        if self._stop - self.rows != 0 or self._stop != self._buf.shape[1]:
            self._buf = self._buf[:, self._stop - self.rows:self._stop].contiguous()
            self._stop = self.rows
        return self._buf.reshape(1, self._buf.shape[0], -1)

    def gather_filters(self, indices: torch.Tensor):
        from .. import _lib as L

        if self._buf is None or not self._batched:
            return
        src = self._live_columns()
        _, b, n = src.shape
        if not src.is_cuda or indices.numel() != b:
            self._buf = self._buf[indices]
            return
        idx = indices.nonzero().reshape(-1) if indices.dtype == torch.bool else indices
        idx = idx.to(device=src.device, dtype=torch.int64).contiguous()
        dst = torch.empty_like(src)
        L.check(L.load().pf_columns_gather(src.data_ptr(), idx.data_ptr(), dst.data_ptr(), n, b, 1, src.element_size(),
                                           L.stream_ptr()), "pf_columns_gather")
        self._buf = dst.reshape(self._buf.shape)

    def exchange_filters(self, other: "MomentLog", mask: torch.Tensor):
        from .. import _lib as L

        if self._buf is None or other._buf is None or not self._batched:
            return
        if self.rows != other.rows:
            raise ValueError(f"cannot exchange histories of different lengths: {self.rows} != {other.rows}")
        dst, src = self._live_columns(), other._live_columns()
        _, b, n = dst.shape
        if not dst.is_cuda or mask.dtype != torch.bool or mask.numel() != b or dst.dtype != src.dtype:
            self._buf[mask] = other._buf[mask]
            return
        m = mask.to(dst.device).contiguous()
        L.check(L.load().pf_columns_exchange(dst.data_ptr(), src.data_ptr(), m.data_ptr(), n, b, 1, dst.element_size(),
                                             L.stream_ptr()), "pf_columns_exchange")


class FilterResult(dict, Generic[TCorrection]):
    def __init__(self, init_state: TCorrection, record_states: BoolOrInt, record_moments: BoolOrInt,
                 _defer_moments: bool = False):
        super().__init__()
        # NB: aliases the initial state's ``_ll`` tensor, exactly like the reference (result.py:34)
        self._loglikelihood = init_state.get_loglikelihood()
        self._moments = MomentLog(_deque_maxlen(record_moments))
        self._states = make_dequeue(maxlen=record_states)
        if _defer_moments:
            # the fused driver: the initial state's moment row arrives with the run's rows (``_extend_fused``); everything
            # else ``append`` does happens here
            self._loglikelihood.add_(init_state.get_loglikelihood())
            self._states.append(init_state)
        else:
            self.append(init_state)

    @staticmethod
    def states_kept(record_states: BoolOrInt) -> Optional[int]:
        """How many states a result keeps (``None`` = all): the reference's deque rule (container.py:10-18)."""
        return _deque_maxlen(record_states)

    # ---- the reference's read interface -------------------------------------------------------------------------
    @property
    def loglikelihood(self) -> torch.Tensor:
        return self._loglikelihood

    @property
    def filter_means(self) -> torch.Tensor:
        """``(timesteps + 1, [batch], latent dim)`` - row 0 is the initial state."""
        return self._moments.means()

    @property
    def filter_variance(self) -> torch.Tensor:
        return self._moments.variances()

    @property
    def states(self) -> List[TCorrection]:
        return list(self._states)

    @property
    def latest_state(self) -> TCorrection:
        return self._states[-1]

    @property
    def tensor_tuples(self) -> TensorContainer:
        """The reference's container view of the moment series (``pyfilter/state.py:19``), built on demand."""
        tc = TensorContainer()
        tc.make_deque("filter_means", self.filter_means.unbind(0) if self._moments.rows else None, maxlen=self._moments.maxlen)
        tc.make_deque("filter_variances", self.filter_variance.unbind(0) if self._moments.rows else None, maxlen=self._moments.maxlen)
        return tc

    # ---- writing ------------------------------------------------------------------------------------------------
    def append(self, state: TCorrection, _ll_accumulated: bool = False):
        """One more state (result.py:119-133): its moments join the log, its log-likelihood the running total
        (``_ll_accumulated``: the fused move that produced the state already added it - the total was its
        ``pf_filter_args.ll_total``)."""
        batched = self._loglikelihood.dim() > 0
        self._moments.append(state.get_mean(), state.get_variance(), batched)
        if not _ll_accumulated:
            self._loglikelihood.add_(state.get_loglikelihood())
        self._states.append(state)
        return self

    def _extend_fused(self, means: torch.Tensor, variances: torch.Tensor, ll_total: torch.Tensor, last_state,
                      states=None):
        """Adopts what a fused run produced: ``(rows, [B], D)`` moment rows (incl. the initial state's when its moments
        were deferred), the run's total log-likelihood, the final state (or, with recorded states, all of them)."""
        if self._moments._buf is None:
            self._moments._adopt_shape(means[0], self._loglikelihood.dim() > 0)
        self._moments.extend(means, variances)
        self._loglikelihood.add_(ll_total)
        if states is not None:
            self._states.extend(states)
        else:
            self._states.append(last_state)
        return self

    # ---- whole-filter moves (SMC^2 / PMMH: result.py:76-117) ------------------------------------------------------
    def exchange(self, other: "FilterResult", mask: torch.Tensor):
        """Overwrites the filters selected by ``mask`` (batch dim) with those of ``other``."""
        from .particle.state import _masked_assign

        _masked_assign(self._loglikelihood, other.loglikelihood, mask)
        self._moments.exchange_filters(other._moments, mask)
        for mine, theirs in zip(self._states, other._states):
            mine.exchange(theirs, mask)
        return self

    def resample(self, indices: torch.Tensor, entire_history: bool = True):
        """Gathers whole filters along the batch dim; ``entire_history=False`` leaves the moment series alone."""
        self._loglikelihood.copy_(self._loglikelihood[indices])
        if entire_history:
            self._moments.gather_filters(indices)
            # Reference quirk, kept: there a recorded state's mean / variance tensors ARE entries of the series (append
            # stores the same tensor objects, result.py:129-130), so gathering the series in place (result.py:111-114)
            # already gathers them - and the state's own resample below gathers them a second time.
            for s in list(self._states)[-self._moments.rows:]:
                hook = getattr(s, "_gather_moments", None)
                if hook is not None:
                    hook(indices)
        for s in self._states:
            s.resample(indices)
        return self

    # ---- wire format (container.py:113-139, result.py:135-154) --------------------------------------------------------
    def state_dict(self):
        tag = f"tensor_deque_{self._moments.maxlen}__"
        series = OrderedDict([(tag + "filter_means", self.filter_means), (tag + "filter_variances", self.filter_variance)])
        return OrderedDict([("tensor_tuples", series), ("state", self.latest_state.state_dict()),
                            ("log_likelihood", self.loglikelihood)])

    def load_state_dict(self, state_dict):
        assert len(self._states) == 1, "Can only handle case when we have 1 state!"
        series = state_dict["tensor_tuples"]
        found = {}
        for key, value in series.items():
            kind, name = key[len("tensor_"):].split("__", 1)
            if kind.startswith("deque_"):
                maxlen = kind.split("_", 1)[1]
                found[name] = (value, None if maxlen == "None" else int(maxlen))
        means, maxlen = found["filter_means"]
        variances, _ = found["filter_variances"]
        self._loglikelihood = state_dict["log_likelihood"]
        self._moments = MomentLog(maxlen)
        if means.numel():
            batched = self._loglikelihood.dim() > 0
            self._moments._adopt_shape(means[0], batched)
            self._moments.extend(means, variances)
        self.latest_state.load_state_dict(state_dict["state"])

    def copy(self) -> "FilterResult":
        return deepcopy(self)

    def __repr__(self):
        return f"FilterResult(ll: {self._loglikelihood!r}, num_observations: {self._moments.rows})"
