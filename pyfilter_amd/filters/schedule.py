"""The move schedule of a filter run - one place for the rule both drivers follow.

A state-space model observed every ``observe_every_step``-th time step (``StateSpaceModel.observe_every_step``) makes
one observation cost ``pad`` propagate-only moves followed by one weighted move (``pyfilter/filters/base.py:204-210``:
*"while the predicted time index is not a multiple of observe_every_step: propagate"*).  An all-NaN observation turns
its weighted move into a propagate-only one as well (``filters/base.py:212``).

* the generic driver (``BaseFilter.filter``) walks :func:`unobserved_moves_before` one observation at a time;
* the fused driver (``ParticleFilter._batch_filter_fused``) asks :func:`expand` for the whole run's schedule and bakes it
  into the launch arguments of ``pf_filter_run`` (``include/pf_amd.h``: ``y`` rows + ``observed`` flags per move).
"""
from typing import List, NamedTuple


def unobserved_moves_before(time_index: int, observe_every_step: int) -> int:
    """Propagate-only moves a filter at ``time_index`` makes before its next weighted move."""
    return (-int(time_index)) % int(observe_every_step)


class Schedule(NamedTuple):
    source: List[int]   # per move: index of the observation it weighs against, -1 for a propagate-only sub-step
    rows: List[int]     # per observation: the move (0-based) that consumed it - its state is the one reported for it

    @property
    def moves(self) -> int:
        return len(self.source)


def expand(time_index: int, observations: int, observe_every_step: int) -> Schedule:
    """The moves of ``observations`` consecutive observations for a filter currently at ``time_index``."""
    source, rows, t = [], [], int(time_index)
    for k in range(observations):
        pad = unobserved_moves_before(t, observe_every_step)
        source.extend([-1] * pad)
        rows.append(len(source))
        source.append(k)
        t += pad + 1
    return Schedule(source, rows)
