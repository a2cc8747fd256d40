"""theta-particles: the parameters of B parallel filters with their priors - the slice of the reference's
``InferenceContext`` / ``PriorBoundParameter`` (``pyfilter/inference/context.py:193-270``, ``parameter.py:79-107``,
``prior.py:47-123``) that SMC^2 and PMMH use.

Every parameter is ONE tensor of shape ``(B, *event)`` that the model holds by reference: the filters of this library
read parameter tensors live (in-place updates re-pack the kernels' parameter rows), so ``unstack_parameters`` /
``exchange`` / ``resample`` write in place and the next filter move sees the new values - nothing is rebuilt."""
import inspect
from collections import OrderedDict
from typing import Dict, Optional

import torch
from torch.distributions import Distribution, Independent, TransformedDistribution
from torch.distributions.constraint_registry import biject_to


def _on_device(d: Distribution, device, dtype) -> Distribution:
    """The same distribution with its parameter tensors on ``device`` (rebuilt from ``arg_constraints``, which names the
    constructor arguments of every standard torch distribution)."""
    if isinstance(d, Independent):  # (its one argument is a distribution)
        return Independent(_on_device(d.base_dist, device, dtype), d.reinterpreted_batch_ndims, validate_args=False)
    args = {}
    for name in d.arg_constraints:
        v = d.__dict__.get(name)
        if v is None and isinstance(inspect.getattr_static(type(d), name, None), property):
            v = getattr(d, name)  # an argument kept in another form (Beta's concentrations live in its Dirichlet)
        if v is None:  # lazily derived alternatives (e.g. MultivariateNormal's precision / covariance forms)
            continue
        args[name] = v.to(device=device, dtype=dtype) if isinstance(v, torch.Tensor) and v.is_floating_point() else v
    try:
        # (no argument validation: torch's ``_validate_sample`` / ``_validate_args`` end in ``if not valid.all(): raise`` - a
        # device -> host round trip per ``log_prob``, i.e. per prior and PMMH move, while the re-filter is running)
        return type(d)(**args, validate_args=False)
    except TypeError:
        return d


def _native_family(d: Distribution, dtype=None):
    """``(PF_PRIOR_* code, a, b)`` when ``d`` is a scalar prior of a family the theta kernels evaluate (``include/pf_amd.h``:
    the family's density and the bijection ``biject_to(d.support)`` that goes with it), else ``None``.  Exact types only: a
    subclass may override ``log_prob`` or ``support``.  ``dtype``: the parameters are rounded to it first - the values the
    device copy of the prior holds (read from the caller's distribution: no device round trip)."""
    from torch.distributions import Beta, Exponential, Gamma, HalfNormal, LogNormal, Normal, Uniform

    table = {Normal: (0, "loc", "scale"), LogNormal: (1, "loc", "scale"), Exponential: (2, "rate", None),
             Gamma: (3, "concentration", "rate"), HalfNormal: (4, "scale", None), Beta: (5, "concentration1", "concentration0"),
             Uniform: (6, "low", "high")}
    row = table.get(type(d))
    if row is None or d.event_shape.numel() != 1 or len(d.event_shape) or d.batch_shape.numel() != 1:
        return None
    vals = []
    for name in row[1:]:
        v = getattr(d, name) if name is not None else 1.0
        v = v.reshape(-1)[0] if isinstance(v, torch.Tensor) else torch.tensor(float(v))
        vals.append(float(v.to(dtype) if dtype is not None and v.is_floating_point() else v))
    return (row[0], vals[0], vals[1])


class Prior:
    """A prior with its bijection to unconstrained space (``prior.py:47-123``)."""

    def __init__(self, distribution: Distribution, device=None, dtype=None):
        self.native = _native_family(distribution, dtype if device is not None else None)
        if device is not None:
            distribution = _on_device(distribution, device, dtype)
        self.distribution = distribution
        self.bijection = biject_to(distribution.support)
        self.unconstrained = TransformedDistribution(distribution, self.bijection.inv, validate_args=False)

    @property
    def numel(self) -> int:
        return max(1, self.distribution.event_shape.numel())

    def get_unconstrained(self, x: torch.Tensor) -> torch.Tensor:
        return self.bijection.inv(x)

    def get_constrained(self, u: torch.Tensor) -> torch.Tensor:
        return self.bijection(u)

    def eval_prior(self, x: torch.Tensor, constrained: bool = True) -> torch.Tensor:
        if constrained:
            return self.distribution.log_prob(x)
        return self.unconstrained.log_prob(self.get_unconstrained(x))


def _sum_event(ladj: torch.Tensor, prior: "Prior") -> torch.Tensor:
    """``log |det J|`` of the bijection summed over the prior's event dimensions the transform treats elementwise."""
    extra = len(prior.distribution.event_shape) - prior.bijection.codomain.event_dim
    return ladj.sum(tuple(range(-extra, 0))) if extra > 0 else ladj


class ThetaParticles:
    """Named parameters of ``batch`` parallel filters.  ``self[name]`` is the tensor to hand to the model."""

    def __init__(self, priors: Dict[str, Distribution], batch: int, device="cuda", dtype=torch.float32, shard=None):
        self.device, self.dtype = torch.device(device), dtype
        self.priors = OrderedDict((k, v if isinstance(v, Prior) else Prior(v, self.device, dtype)) for k, v in priors.items())
        self.batch_shape = torch.Size([batch])
        self.shard = shard  # pyfilter_amd.distributed.Shard or None: `batch` is this rank's block of the theta-particles
        self._values: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        # Derived quantities of the CURRENT values, kept across calls: "u" the stacked unconstrained parameters (B, P), "prior_u"
        # the summed log priors in unconstrained space (B,).  A rejuvenation asks for each of them several times (the proposal
        # fit, the reverse kernel, the acceptance ratio) and every evaluation is ~10 / ~45 small launches through
        # torch.distributions' transform machinery; the whole-filter moves below carry them along (an index / a select)
        # instead of dropping them.  Every method that writes the values maintains or clears the cache - the tensors are
        # handed to the model by reference, but nothing outside this class writes them.
        self._cache: Dict[str, torch.Tensor] = {}
        # Scalar parameters (every prior without an event shape - the usual case) live side by side in ONE ``(P + 1, B)``
        # buffer: ``self[name]`` is row k (a contiguous ``(B,)`` tensor), the last row carries ``_cache["prior_u"]``.  A whole-
        # filter move (``resample`` / ``exchange``) of a rejuvenation is then two launches / one instead of two per parameter
        # and cached quantity (3 parameters: 8 -> 3 and 8 -> 2) - and one collective instead of P + 2 when sharded.
        self._buf: Optional[torch.Tensor] = None

    def _store(self, fresh: "OrderedDict[str, torch.Tensor]"):
        """First values: allocates the storage (``_buf`` rows when every parameter is a scalar per filter)."""
        b = self.batch_shape[0]
        if all(v.dim() == 1 and v.shape[0] == b for v in fresh.values()) and len(fresh) > 0:
            self._buf = torch.empty((len(fresh) + 1, b), device=self.device, dtype=self.dtype)
            for k, (name, v) in enumerate(fresh.items()):
                self._buf[k].copy_(v)
                self._values[name] = self._buf[k]
        else:
            for name, v in fresh.items():
                self._values[name] = v.contiguous()

    def _prior_row(self) -> Optional[torch.Tensor]:
        """Where the summed log prior of the unconstrained values lives when the parameters are packed (else ``None``)."""
        return None if self._buf is None else self._buf[-1]

    # ---- values ---------------------------------------------------------------------------------------------------
    def initialize_parameters(self, generator: Optional[torch.Generator] = None):
        """Draws every parameter from its prior (``context.initialize_parameters``): ``(B, *event)`` per name.  With a (CPU)
        ``generator`` the draws are reproducible - inverse CDF of its uniforms where the prior has one - and, sharded, every
        rank draws the same global set and keeps its block."""
        total = self.shard.total if self.shard is not None else self.batch_shape[0]
        fresh = OrderedDict()
        for name, prior in self.priors.items():
            d = prior.distribution
            shape = torch.Size([total])
            v = None
            if generator is not None:
                try:
                    u = torch.rand(shape + d.event_shape, generator=generator, dtype=torch.float64).clamp(1e-12, 1 - 1e-12)
                    v = d.icdf(u.to(device=self.device, dtype=self.dtype))
                except NotImplementedError:
                    v = None
            if v is None and generator is not None:
                # no inverse CDF (Beta, Gamma, event-shaped priors ...): the distribution's own sampler under a forked
                # global stream seeded from the generator - still the same draws on every rank
                word = int(torch.randint(0, 2 ** 62, (), generator=generator))
                devs = [self.device] if self.device.type == "cuda" else []
                with torch.random.fork_rng(devices=devs):
                    torch.manual_seed(word)
                    v = d.sample(shape)
            if v is None:
                v = d.sample(shape)
            if self.shard is not None:
                v = self.shard.slice(v)
            fresh[name] = v.to(device=self.device, dtype=self.dtype)
        if self._values:
            for name, v in fresh.items():
                self._values[name].copy_(v)
        else:
            self._store(fresh)
        self._cache.clear()
        return self

    def __getitem__(self, name: str) -> torch.Tensor:
        return self._values[name]

    def native_priors(self):
        """The priors as the theta kernels take them (``_lib.PfThetaPriors``; ``ops.theta_propose``) - or ``None``: a prior
        of another family / with an event shape, more than ``PF_THETA_MAXP`` parameters, a CPU run."""
        from ..hints import HINTS

        if not HINTS.theta_kernels or len(self._values) != len(self.priors):
            return None
        if "native" not in self.__dict__:
            from .. import _lib

            ok = (self.device.type == "cuda" and 0 < len(self.priors) <= _lib.THETA_MAXP and all(p.native is not None for p in self.priors.values()) and
                  all(v.dim() == 1 and v.is_contiguous() for v in self._values.values()) and
                  self.dtype in (torch.float32, torch.float64))
            packed = None
            if ok:
                packed = _lib.PfThetaPriors()
                packed.P = len(self.priors)
                for i, p in enumerate(self.priors.values()):
                    packed.kind[i], packed.a[i], packed.b[i] = p.native
            self.native = packed
        return self.native

    def adopt_proposal(self, u: torch.Tensor, prior_u: torch.Tensor):
        """The value tensors were just written by ``ops.theta_propose`` from the unconstrained ``u (B, P)``: what is known
        about them (``u`` itself, their summed log prior) replaces whatever was known about the old values."""
        self._cache = {"u": u}
        self._keep_prior(prior_u)

    def _keep_prior(self, prior_u: torch.Tensor):
        row = self._prior_row()
        if row is not None and prior_u.data_ptr() != row.data_ptr():
            row.copy_(prior_u)
        self._cache["prior_u"] = prior_u if row is None else row

    def names(self):
        return list(self.priors)

    def like(self) -> "ThetaParticles":
        """An independent set with the same priors / shape (the proposal's parameters: ``context.make_new``)."""
        other = ThetaParticles(self.priors, self.batch_shape[0], self.device, self.dtype, self.shard)
        if self._buf is not None:
            other._buf = self._buf.clone()
            for k, name in enumerate(self._values):
                other._values[name] = other._buf[k]
        else:
            for k, v in self._values.items():
                other._values[k] = v.clone()
        if "native" in self.__dict__:
            other.native = self.native
        other._cache = dict(self._cache)  # (same values: the derived quantities hold - "u" is a tensor nobody writes in place,
        if "prior_u" in other._cache and other._buf is not None:  # the log prior travels in the buffer's last row)
            other._cache["prior_u"] = other._buf[-1]
        return other

    # ---- stacked views (context.py:193-243) -----------------------------------------------------------------------
    def stack_parameters(self, constrained: bool = True) -> torch.Tensor:
        """``(B, P)``: the parameters side by side, flattened per filter."""
        if not constrained and "u" in self._cache:
            return self._cache["u"]
        cols = []
        for name, prior in self.priors.items():
            v = self._values[name]
            cols.append((v if constrained else prior.get_unconstrained(v)).reshape(self.batch_shape[0], -1))
        out = torch.cat(cols, dim=-1)
        if not constrained:
            self._cache["u"] = out
            # (a bijection may change the event size - simplex / stick-breaking: the unconstrained columns of a prior are then
            # NOT its constrained numel; the widths are what slices ``u`` back apart)
            self._u_widths = [c.shape[-1] for c in cols]
        return out

    def _unconstrained_widths(self):
        """Columns every prior occupies in the stacked UNCONSTRAINED tensor (``stack_parameters(False)``)."""
        w = getattr(self, "_u_widths", None)
        if w is None:
            w = self._u_widths = [prior.get_unconstrained(self._values[name]).reshape(self.batch_shape[0], -1).shape[-1]
                                  for name, prior in self.priors.items()]
        return w

    def unstack_parameters(self, x: torch.Tensor, constrained: bool = True):
        """Writes ``x (B, P)`` back into the parameter tensors - in place."""
        at = 0
        widths = [] if constrained else self._unconstrained_widths()
        for i, (name, prior) in enumerate(self.priors.items()):
            v = self._values[name]
            k = max(1, v[0].numel()) if constrained else widths[i]
            part = x[..., at:at + k]
            if constrained:
                v.copy_(part.reshape(v.shape))
            else:
                same = k == max(1, v[0].numel())
                v.copy_(prior.get_constrained(part.reshape(v.shape) if same else part.reshape(v.shape[:1] + (k,))).reshape(v.shape))
            at += k
        self._cache.clear()
        if not constrained and x.dim() == 2 and x.shape[0] == self.batch_shape[0] and at == x.shape[1]:
            # the unconstrained values a later stack / prior evaluation starts from - a COPY in the particles' own type: ``x`` is
            # the caller's tensor (a proposal may reuse its sample buffer in place)
            self._cache["u"] = x.detach().to(self.dtype).clone()

    def eval_priors(self, constrained: bool = True) -> torch.Tensor:
        """``(B,)`` sum of the log priors (``context.py:245-253``).  Unconstrained: the density of ``u = bijection^-1(x)``,
        ``log p(x) + log |dx / du|`` - evaluated from the stacked ``u`` when it is at hand (no inverse transforms), kept
        until the values change."""
        if constrained:
            return sum(p.eval_prior(self._values[n], True) for n, p in self.priors.items())
        if "prior_u" in self._cache:
            return self._cache["prior_u"]
        u_all, at, total = self._cache.get("u"), 0, 0.0
        widths = self._unconstrained_widths() if u_all is not None else None
        for i, (name, prior) in enumerate(self.priors.items()):
            v = self._values[name]
            k = widths[i] if widths is not None else 0
            if u_all is None or k != max(1, v[0].numel()):  # (a size-changing bijection: evaluated on the prior's own tensors)
                lp = prior.eval_prior(v, False)
            else:
                u = u_all[..., at:at + k].reshape(v.shape)
                lp = prior.distribution.log_prob(v) + _sum_event(prior.bijection.log_abs_det_jacobian(u, v), prior)
            total = total + lp
            at += k
        self._keep_prior(total)
        return self._cache["prior_u"]

    # ---- whole-filter moves ---------------------------------------------------------------------------------------
    def exchange(self, other: "ThetaParticles", mask: torch.Tensor):
        packed = self._buf is not None and other._buf is not None and other._buf.shape == self._buf.shape
        if packed:
            torch.where(mask, other._buf, self._buf, out=self._buf)  # (every parameter and the log-prior row: one launch)
        else:
            for name, v in self._values.items():
                m = mask.reshape(mask.shape + (1,) * (v.dim() - 1))
                v.copy_(torch.where(m, other._values[name], v))
        kept = {}
        for key in ("u", "prior_u"):  # the derived quantities move with the values
            if key in self._cache and key in other._cache:
                mine = self._cache[key]
                if key == "prior_u" and packed:
                    kept[key] = mine  # (the buffer's last row: moved above)
                else:
                    kept[key] = torch.where(mask.reshape(mask.shape + (1,) * (mine.dim() - 1)), other._cache[key], mine)
        self._cache = {}
        if "u" in kept:
            self._cache["u"] = kept["u"]
        if "prior_u" in kept:
            self._keep_prior(kept["prior_u"])

    def resample(self, indices: torch.Tensor, route=None):
        """``theta <- theta[indices]`` in place.  Sharded: ``indices`` are GLOBAL ancestors of this rank's positions
        (``route``: the exchange plan of that gather when the caller already built it, ``distributed.Route``)."""
        if self.shard is not None and self.shard.collective and route is None:
            route = self.shard.route(indices)
        moved = (lambda c: c[indices]) if route is None else route.take
        if self._buf is not None:
            self._buf.copy_(self._buf.index_select(1, indices) if route is None else route.take(self._buf, dim=1))
        else:
            for name, v in self._values.items():
                v.copy_(moved(v))
        kept = dict(self._cache)
        self._cache = {}
        if "u" in kept:
            self._cache["u"] = moved(kept["u"])
        if "prior_u" in kept:
            self._keep_prior(kept["prior_u"] if self._buf is not None else moved(kept["prior_u"]))

    def state_dict(self):
        return OrderedDict((k, v.clone()) for k, v in self._values.items())

    def load_state_dict(self, sd):
        for k, v in sd.items():
            self._values[k].copy_(v)
        self._cache.clear()
