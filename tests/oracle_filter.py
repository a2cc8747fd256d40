"""TEST INFRASTRUCTURE - a CPU stand-in for a ``pyfilter_amd`` particle filter, backed by the oracle
(``oracle/cpu_ref.py``), with the exact interface ``pyfilter_amd.inference`` drives: ``set_batch_shape``,
``initialize_model``, ``initialize``, ``initialize_with_result``, ``filter``, ``batch_filter``, ``copy``,
``increase_particles``.  It lets the multi-process (gloo) tests run the product's SMC^2 / PMMH / sharding code on CPU.

Every random draw is keyed by (global theta-particle, time index, run), so a sharded run reproduces the unsharded one
exactly - which is what those tests assert."""
import torch

from oracle import cpu_ref, models as M
from pyfilter_amd.filters.particle.state import ParticleFilterCorrection
from pyfilter_amd.filters.result import FilterResult
from pyfilter_amd.timeseries import TimeseriesState


class OracleState(ParticleFilterCorrection):
    """The product's state class with the two batch-dim moves done by torch indexing (the product's run in HIP kernels)."""

    def resample(self, indices):
        ts = self.timeseries_state
        self["_x"] = ts.copy(values=ts.value[:, indices])
        self["_w"] = self["_w"][:, indices]
        self["_prev_inds"] = self["_prev_inds"][:, indices]
        for k in ("_ll", "_mean", "_var"):
            self[k] = self[k][indices]

    def exchange(self, other, mask):
        ts = self.timeseries_state
        v = ts.value.clone()
        v[:, mask] = other.timeseries_state.value[:, mask]
        self["_x"] = ts.copy(values=v)
        for k in ("_w", "_prev_inds"):
            t = self[k].clone()
            t[:, mask] = other[k][:, mask]
            self[k] = t
        for k in ("_ll", "_mean", "_var"):
            t = self[k].clone()
            t[mask] = other[k][mask]
            self[k] = t


def _draws(columns, t, run, n, kind):
    """(n, B_local) draws, column c of time t of run `run` from its own seeded stream."""
    out = []
    for c in columns:
        g = torch.Generator().manual_seed(1_000_003 * int(c) + 7919 * int(t) + 104_729 * int(run) + kind)
        out.append(torch.randn(n, generator=g, dtype=torch.float64) if kind == 0 else torch.rand((), generator=g, dtype=torch.float64).expand(1))
    return torch.stack(out, dim=1) if kind == 0 else torch.cat(out)


class OracleAPF:
    """APF + Bootstrap on the OU model of ``tests/inference/models.py`` (theta = (kappa, gamma, sigma) per filter)."""

    runs = 0  # class-wide run counter: advanced identically in every process

    PROPOSAL = "bootstrap"
    STATIONARY_INIT = False  # True: x0 ~ N(gamma, sigma / sqrt(2 kappa)) - the OU model of tests/inference/models.py

    def __init__(self, model_builder, particles, columns=None, seed=0, **_ignored):
        self._builder = model_builder
        self._n = particles
        self._z_tape = self._u_tape = self._z0 = None
        self._b = 1
        self._columns = columns  # global ids of this rank's theta-particles (set by the test through `shard`)
        self._theta = None
        self.shard = None

    # ---- the interface pyfilter_amd.inference uses ----------------------------------------------------------------------
    @property
    def batch_shape(self):
        return torch.Size([self._b])

    @property
    def particles(self):
        return torch.Size([self._n, self._b])

    def set_batch_shape(self, shape):
        self._b = shape[0]

    def initialize_model(self, theta):
        self._theta = theta
        self.shard = theta.shard

    def increase_particles(self, factor):
        self._n = int(self._n * factor)

    def copy(self):
        f = type(self)(self._builder, self._n)
        f._b = self._b
        return f

    @property
    def _base_particles(self):
        return torch.Size([self._n])

    def set_tape(self, z=None, u=None, z0=None):
        """The product's parity-mode interface (``ParticleFilter.set_tape``): injected draws instead of the keyed streams."""
        self._z_tape, self._u_tape, self._z0 = z, u, z0

    def _cols(self):
        return range(self.shard.lo, self.shard.hi) if self.shard is not None else range(self._b)

    def _spec(self):
        t = self._theta
        k, g, s = t["kappa"].double(), t["gamma"].double(), t["sigma"].double()
        init = (g, s / torch.sqrt(2.0 * k)) if self.STATIONARY_INIT else (0.0, 0.1)
        return M.ModelSpec(M.HID_OU, (k, g, s), 0, 1.0, init, M.OBS_LINEAR, (1.0, 0.0, 0.05), 0)

    def initialize(self):
        OracleAPF.runs += 1
        self._run = OracleAPF.runs
        z0 = self._z0.double() if self._z0 is not None else _draws(self._cols(), -1, self._run, self._n, 0)
        x0 = M.initial_sample(self._spec(), z0)
        w = torch.zeros(self._n, self._b, dtype=torch.float64)
        idx = torch.arange(self._n).unsqueeze(-1).expand(self._n, self._b)
        return self._state(0, x0, w, torch.zeros(self._b, dtype=torch.float64), idx)

    def _state(self, t, x, w, ll, idx):
        mean, var = cpu_ref.get_filter_mean_and_variance(x, cpu_ref.normalize(w.clone()), False)
        return OracleState(TimeseriesState(t, x, torch.Size([])), w, ll, idx, _moments=(mean, var))

    def initialize_with_result(self, state=None):
        return FilterResult(state if state is not None else self.initialize(), False, True)

    def filter(self, y, state, result=None):
        t = int(state.timeseries_state.time_index)
        run = getattr(self, "_run", 0)
        z = self._z_tape[t].double() if self._z_tape is not None else _draws(self._cols(), t, run, self._n, 0)
        u = self._u_tape[t].double() if self._u_tape is not None else _draws(self._cols(), t, run, self._n, 1)
        x, w, ll, idx = cpu_ref.apf_step(self._spec(), self.PROPOSAL, y.double(), state.timeseries_state.value,
                                         state.weights.clone(), z, u)
        new = self._state(t + 1, x, w, ll, idx)
        if result is not None:
            result.append(new)
        return new

    def filter_block(self, y, state, observed=None, replay=None, per_step=False, defer_status=False):
        """The product's look-ahead interface (``ParticleFilter.filter_block``): the moves are keyed by (column, time, run),
        so a cut replay repeats them exactly.  (``per_step`` / ``defer_status``: kernel-route matters of the HIP filters -
        nothing to watch here, ``result.status`` stays absent.)"""
        start = state._restarted()
        result = self.initialize_with_result(start)
        lls, s = [], start
        for yt in y:
            s = self.filter(yt, s, result=result)
            lls.append(s.get_loglikelihood())
        return result, torch.stack(lls), None

    def batch_filter(self, y, bar=False, init_state=None):
        state = init_state if init_state is not None else self.initialize()
        result = self.initialize_with_result(state)
        for yt in y:
            state = self.filter(yt, state, result=result)
        return result
