"""Proposal base class (``pyfilter/filters/particle/proposals/base.py:12-92``).

A proposal evaluates the model in one of two ways:

* the model has a built-in kernel kind (``ssm.kernel_kind``): one HIP elementwise kernel
  (``pf_sample_and_weight`` / ``pf_pre_weight``) - and, inside ``batch_filter``, the fused one-kernel step;
* otherwise the reference's own route: the user's callables evaluated with PyTorch-ROCm ops.
"""
from abc import ABC
from typing import Callable, Optional, Tuple

import torch

from .... import _lib as L
from .... import ops
from ....timeseries import AffineProcess, StateSpaceModel, StructuralStochasticProcess, TimeseriesState


def _affine_pre_weight(mod: AffineProcess, state: TimeseriesState) -> TimeseriesState:
    """Default APF pre-weight state: the deterministic one-step mean (pre_weight_funcs.py:9-11)."""
    loc, _ = mod.mean_scale(state)
    return state.propagate_from(values=loc)


def _missing(mod, state):
    raise Exception("You didn't pass a custom function, and couldn't find a suitable pre-defined one!")


class KernelContext:
    """What a proposal needs to launch the built-in model kernels for one filter: the kind, the packed per-column
    parameter rows, the Philox seed and the optional draw tapes (parity mode)."""

    def __init__(self, kind, params: torch.Tensor, seed: int, batched: bool, has_event: bool):
        self.kind = kind
        self.params = params
        self.seed = seed
        self.batched = batched
        self.has_event = has_event
        self.z_tape: Optional[torch.Tensor] = None  # (T, D, B, N)
        self.u_tape: Optional[torch.Tensor] = None  # (T, B)

    def z_for(self, step: int) -> Optional[torch.Tensor]:
        return None if self.z_tape is None else self.z_tape[step]

    def u_for(self, step: int) -> Optional[torch.Tensor]:
        return None if self.u_tape is None else self.u_tape[step]


class Proposal(ABC):
    _KERNEL_PROPOSAL = None  # PF_PROP_* when the HIP kernels implement this proposal

    def __init__(self, pre_weight_func: Callable[[StructuralStochasticProcess, TimeseriesState], TimeseriesState] = None):
        super().__init__()
        self._model: StateSpaceModel = None
        self._pre_weight_func = pre_weight_func
        self._custom_pre_weight = pre_weight_func is not None
        self._ctx: Optional[KernelContext] = None

    def set_model(self, model: StateSpaceModel):
        self._model = model
        if self._pre_weight_func is None:
            self._pre_weight_func = _affine_pre_weight if isinstance(model.hidden, AffineProcess) else _missing
        return self

    def _set_context(self, ctx: Optional[KernelContext]):
        self._ctx = ctx
        return self

    @property
    def uses_kernels(self) -> bool:
        # (a user-defined affine process has no stand-alone model kernel: its callables run as torch ops on this route; the
        # fused single step takes its (loc, scale) planes)
        return (self._ctx is not None and self._KERNEL_PROPOSAL is not None and not self._custom_pre_weight
                and not self._ctx.kind.is_user)

    # -- kernel route ------------------------------------------------------------------------------------------
    def _kernel_sample_and_weight(self, y, x: TimeseriesState, weigh=True):
        c = self._ctx
        step = int(x.time_index)
        soa = ops.to_soa(x.value, c.batched, c.has_event)
        x_out, w_out = ops.sample_and_weight_soa(
            c.kind, c.params, self._KERNEL_PROPOSAL, soa, y, c.z_for(step), c.seed, step, weigh=weigh
        )
        new_x = x.propagate_from(values=ops.from_soa(x_out, c.batched, c.has_event))
        return new_x, (ops.from_cols(w_out, c.batched) if weigh else None)

    def _kernel_pre_weight(self, y, x: TimeseriesState):
        c = self._ctx
        soa = ops.to_soa(x.value, c.batched, c.has_event)
        return ops.from_cols(ops.pre_weight_soa(c.kind, c.params, self._KERNEL_PROPOSAL, soa, y), c.batched)

    def _propagate(self, x: TimeseriesState) -> TimeseriesState:
        """``model.hidden.propagate`` (no weighting) for unobserved steps."""
        if self.uses_kernels:
            return self._kernel_sample_and_weight(None, x, weigh=False)[0]
        return self._model.hidden.propagate(x)

    # -- reference API -----------------------------------------------------------------------------------------
    def _weight_with_kernel(self, y, x_dist, x_new: TimeseriesState, kernel) -> torch.Tensor:
        y_dist = self._model.build_density(x_new)
        return y_dist.log_prob(y) + x_dist.log_prob(x_new.value) - kernel.log_prob(x_new.value)

    def sample_and_weight(self, y: torch.Tensor, prediction) -> Tuple[TimeseriesState, torch.Tensor]:
        raise NotImplementedError()

    def pre_weight(self, y: torch.Tensor, x: TimeseriesState) -> torch.Tensor:
        """``log p(y_t | pre-weight state)`` used by the APF's first stage (base.py:69-85)."""
        if self.uses_kernels:
            return self._kernel_pre_weight(y, x)
        new_state = self._pre_weight_func(self._model.hidden, x)
        return self._model.build_density(new_state).log_prob(y)

    def copy(self) -> "Proposal":
        raise NotImplementedError()
