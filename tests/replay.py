"""TEST INFRASTRUCTURE - replays the event logs of ``tests/golden/inference_*.npz`` (written by
``oracle/make_golden_inference.py`` from the unmodified reference's SMC^2 / PMMH code) through the PRODUCT's
``pyfilter_amd.inference``: every random number the reference consumed is handed to the product at the same point of
the algorithm - a product that draws in a different order, or a different number of times, fails on the event kinds -
and every quantity the reference computed is compared on the way.

Two filters can sit underneath: the HIP filters (``-m gpu``: the product end to end) and an oracle-backed CPU stand-in
(``-m "not gpu"``: the product's theta-level / host code on CPU, the particle filter being ``oracle/cpu_ref.py``)."""
import os
import re

import numpy as np
import torch

from pyfilter_amd.inference.pmmh import ThetaDraws

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_events(name):
    """``[(kind, {field: tensor | str})]`` in the order the reference produced them."""
    events = {}
    with np.load(os.path.join(GOLDEN, f"{name}.npz")) as f:
        for key in f.files:
            m = re.fullmatch(r"e(\d+)::(\w+)::(\w+)", key)
            k, kind, field = int(m.group(1)), m.group(2), m.group(3)
            v = f[key]
            events.setdefault(k, (kind, {}))[1][field] = torch.from_numpy(v) if v.dtype.kind in "fiub" else v
    return [events[k] for k in sorted(events)]


class Cursor:
    """The shared read position in the event log."""

    def __init__(self, events):
        self.events, self.at = events, 0

    def peek(self):
        return self.events[self.at][0] if self.at < len(self.events) else None

    def take(self, kind):
        got, fields = self.events[self.at]
        assert got == kind, f"event {self.at}: the product asks for '{kind}' where the reference did '{got}'"
        self.at += 1
        return fields


class ReplayDraws(ThetaDraws):
    """The theta-level draws in the reference's order: the resampling uniform of ``update`` (kernels/mh.py:53), the
    proposal's standard normals (mcmc/utils.py:48), the acceptance uniforms (mcmc/utils.py:69)."""

    def __init__(self, cursor: Cursor):
        self.cursor = cursor
        self.generator = None

    def uniform(self, shape):
        shape = tuple(shape)
        if shape == ():
            return self.cursor.take("rejuvenate")["u"].double().reshape(())
        u = self.cursor.take("pmmh_accept")["u"].double()
        assert tuple(u.shape) == shape, (u.shape, shape)
        return u

    def normal(self, shape):
        eps = self.cursor.take("pmmh_draw")["eps"].double()
        assert tuple(eps.shape) == tuple(shape), (eps.shape, shape)
        return eps


def taped(base):
    """A filter class whose every run takes its draws from the event log: ``initialize`` <- ``init``, an online
    ``filter()`` move <- ``move``, a whole ``batch_filter`` <- ``run`` (copies share the cursor)."""

    class Taped(base):
        cursor: Cursor = None

        def initialize(self):
            if getattr(self, "_in_run", False):
                return super().initialize()
            self.set_tape(z0=self.cursor.take("init")["z0"])
            return super().initialize()

        def filter(self, y, state, result=None):
            if getattr(self, "_in_run", False):
                return super().filter(y, state, result=result)
            ev = self.cursor.take("move")
            t = int(state.timeseries_state.time_index)
            z = torch.zeros((t + 1,) + tuple(ev["z"].shape), dtype=ev["z"].dtype)  # (tapes are indexed by the time index)
            u = torch.zeros((t + 1,) + tuple(ev["u"].shape), dtype=ev["u"].dtype)
            z[t], u[t] = ev["z"], ev["u"]
            self.set_tape(z=z, u=u)
            new = super().filter(y, state, result=result)
            # (a copy: a rejuvenation triggered by this move resamples / exchanges the state's tensors in place)
            self.last_move_ll = new.get_loglikelihood().clone()
            return new

        def batch_filter(self, y, bar=False, init_state=None):
            ev = self.cursor.take("run")
            assert int(ev["n"]) == self._base_particles[0] and int(ev["t"]) == y.shape[0], (ev["n"], ev["t"], y.shape)
            self.set_tape(z=ev["z"], u=ev["u"], z0=ev["z0"])
            self._in_run = True
            try:
                res = super().batch_filter(y, bar=bar, init_state=init_state)
            finally:
                self._in_run = False
            self.last_run_ll = (res.loglikelihood, ev["ll"])
            return res

        def filter_block(self, *a, **k):  # the replay is observation by observation, like the reference
            return None

    return Taped


def close(a, b, what, rtol=1e-8, atol=1e-10):
    torch.testing.assert_close(torch.as_tensor(a).detach().cpu().double().reshape(-1),
                               torch.as_tensor(b).detach().cpu().double().reshape(-1), rtol=rtol, atol=atol, msg=lambda m: f"{what}: {m}")


def compare_update(trace, events, lo, hi, what):
    """The product's trace of one ``ParticleMetropolisHastings.update`` against the reference's events ``[lo, hi)``."""
    ref = [(k, f) for k, f in events[lo:hi] if k in ("rejuvenate", "pmmh_draw", "pmmh_accept")]
    moves = [t for t in trace if t["kind"] == "pmmh"]
    rj = [t for t in trace if t["kind"] == "rejuvenate"]
    assert len(rj) == 1 and ref[0][0] == "rejuvenate", what
    assert torch.equal(rj[0]["indices"].cpu(), ref[0][1]["indices"]), f"{what}: theta ancestors"
    close(rj[0]["kernel"].loc, ref[0][1]["kernel_mean"], f"{what}: proposal mean")
    close(rj[0]["kernel"].scale_tril, ref[0][1]["kernel_scale_tril"], f"{what}: proposal scale_tril")
    draws = [f for k, f in ref if k == "pmmh_draw"]
    accepts = [f for k, f in ref if k == "pmmh_accept"]
    assert len(moves) == len(draws) == len(accepts), (what, len(moves), len(draws))
    for i, (mv, dr, ac) in enumerate(zip(moves, draws, accepts)):
        close(mv["rvs"], dr["rvs"], f"{what} move {i}: theta*")
        close(mv["log_acc"], ac["log_acc"], f"{what} move {i}: log acceptance probability", rtol=1e-7, atol=1e-7)
        assert torch.equal(mv["accepted"].cpu(), ac["accepted"]), f"{what} move {i}: accepted mask"
        close(mv["new_kernel"].loc, ac["new_kernel_loc"], f"{what} move {i}: reverse kernel mean")
