#!/usr/bin/env python
"""Wall time of the segments of one SMC2.step() observation (perf_counter around them, no profiler):
python tools/smc2_step_segments.py [n_theta] [n_state] [T]"""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _env  # noqa: E402

_env.setup()


def main():
    from torch.distributions import Exponential, LogNormal, Normal

    from pyfilter_amd import _lib as L
    from pyfilter_amd import ops
    from pyfilter_amd import timeseries as ts
    from pyfilter_amd.filters.particle import APF, proposals
    from pyfilter_amd.inference import SMC2
    from pyfilter_amd.timeseries import models

    n_theta = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    n_state = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    t_len = int(sys.argv[3]) if len(sys.argv) > 3 else 500
    device, dtype = torch.device("cuda"), torch.float32
    g = torch.Generator().manual_seed(123)
    x, ys = 0.0, []
    for _ in range(t_len):
        x = x * math.exp(-0.025) + 0.05 * math.sqrt((1 - math.exp(-0.05)) / 0.05) * torch.randn((), generator=g).item()
        ys.append(x + 0.05 * torch.randn((), generator=g).item())
    y = torch.tensor(ys, dtype=dtype, device=device)
    priors = {"kappa": Exponential(10.0), "gamma": Normal(0.0, 1.0), "sigma": LogNormal(-2.0, 1.0)}
    obs_a, obs_s = torch.tensor(1.0, dtype=dtype, device=device), torch.tensor(0.05, dtype=dtype, device=device)

    def build(theta):
        return ts.LinearStateSpaceModel(models.OrnsteinUhlenbeck(theta["kappa"], theta["gamma"], theta["sigma"], dt=1.0), (obs_a, obs_s))

    acc = {}

    def timed(obj, name, key):
        f = getattr(obj, name)

        def w(*a, **k):
            t1 = time.perf_counter()
            try:
                return f(*a, **k)
            finally:
                acc[key] = acc.get(key, 0.0) + time.perf_counter() - t1
        setattr(obj, name, w)

    lib = L.load()
    real_run = lib.pf_filter_run

    for seed in range(3):
        acc.clear()
        filt = APF(build, n_state, proposal=proposals.LinearGaussianObservations(), seed=2024 + seed)
        alg = SMC2(filt, n_theta, priors, threshold=0.2, device=device, dtype=dtype, seed=seed)
        state = alg.initialize()
        timed(filt, "_filter_fused_single", "filter: _filter_fused_single")
        timed(filt, "_ensure_context", "   of it: _ensure_context")
        timed(alg._kernel, "update", "rejuvenations (update)")

        def run_c(*a):
            t1 = time.perf_counter()
            r = real_run(*a)
            acc["   of it: pf_filter_run (C)"] = acc.get("   of it: pf_filter_run (C)", 0.0) + time.perf_counter() - t1
            return r
        lib.pf_filter_run = run_c
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for yt in y:
            ta = time.perf_counter()
            state.append_data(yt)
            fs = alg.filter.filter(yt, state.filter_state.latest_state, result=state.filter_state)
            tb = time.perf_counter()
            slot = alg.__dict__.get("_host_slot")
            if slot is None:
                slot = alg._host_slot = ops.HostSlot()
            on_host = state.append(fs, slot)
            tc = time.perf_counter()
            ess, finite = slot.wait() if on_host else state.stats.tolist()
            td = time.perf_counter()
            if ess < alg._threshold * alg.particles[0] or not finite:
                state = alg._kernel.update(alg.theta, alg.filter, state, generator=alg._gen)
            state.current_iteration += 1
            acc["filter() incl. result.append"] = acc.get("filter() incl. result.append", 0.0) + tb - ta
            acc["state.append (theta weights + ESS)"] = acc.get("state.append (theta weights + ESS)", 0.0) + tc - tb
            acc["host has (ESS, finite) (slot.wait)  "] = acc.get("host has (ESS, finite) (slot.wait)  ", 0.0) + td - tc
        torch.cuda.synchronize()
        total = time.perf_counter() - t0
        lib.pf_filter_run = real_run
    print(f"step() loop at {n_theta} x {n_state}, T = {t_len}: {1e3 * total:.1f} ms; per observation (us):")
    for k, v in acc.items():
        print(f"   {k:40s} {1e6 * v / t_len:7.2f}" + ("   (ms in total: %.2f)" % (1e3 * v) if "rejuv" in k else ""))


if __name__ == "__main__":
    main()
