#!/usr/bin/env python
"""Where the DEVICE idles during an SMC^2 fit (development tool).  Two modes:
  python tools/smc2_gpu_gaps.py run [n_theta] [n_state]   - three fits separated by 50 ms sleeps (run it under rocprofv3 --kernel-trace)
  python tools/smc2_gpu_gaps.py read <kernel_trace.csv>   - the last fit of that trace: span, busy time, the idle gaps and what ran around them"""
import csv
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def run(n_theta, n_state):
    import torch
    from torch.distributions import Exponential, LogNormal, Normal

    import _env

    _env.setup()
    from pyfilter_amd import timeseries as ts
    from pyfilter_amd.filters.particle import APF, proposals
    from pyfilter_amd.inference import SMC2
    from pyfilter_amd.timeseries import models

    device, dtype, t_len = torch.device("cuda"), torch.float32, 500
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

    for rep in range(3):
        filt = APF(build, n_state, proposal=proposals.LinearGaussianObservations(), seed=2024 + 3)
        alg = SMC2(filt, n_theta, priors, threshold=0.2, device=device, dtype=dtype, seed=3)
        torch.cuda.synchronize()
        time.sleep(0.05)
        t0 = time.perf_counter()
        alg.fit(y)
        torch.cuda.synchronize()
        print(f"fit {rep}: {1e3 * (time.perf_counter() - t0):.2f} ms, PMMH moves {len(alg._kernel.acceptance_history)}")


def read(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # the last fit: everything after the last idle gap of more than 30 ms
    cut = 0
    for i in range(1, len(rows)):
        if rows[i][0] - rows[i - 1][1] > 30_000_000:
            cut = i
    rows = rows[cut:]
    span = rows[-1][1] - rows[0][0]
    busy = sum(e - s for s, e, _ in rows)
    print(f"last fit: {len(rows)} kernels, span {span / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, idle {(span - busy) / 1e6:.2f} ms")
    by = {}
    for s, e, n in rows:
        k = n.split("(")[0][:70]
        by[k] = (by.get(k, (0, 0))[0] + 1, by.get(k, (0, 0))[1] + e - s)
    for k, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"   {t / 1e3:9.1f} us  {c:5d} x  {k}")
    gaps = sorted(((rows[i][0] - rows[i - 1][1], i) for i in range(1, len(rows))), reverse=True)
    hist = {}
    for gap, _ in gaps:
        b = "<5us" if gap < 5000 else "<20us" if gap < 20000 else "<100us" if gap < 100000 else ">=100us"
        hist[b] = (hist.get(b, (0, 0))[0] + 1, hist.get(b, (0, 0))[1] + gap)
    print("   idle by gap size:", {k: f"{c} gaps, {t / 1e6:.2f} ms" for k, (c, t) in hist.items()})
    print("   the 14 longest gaps (us): before <- after")
    for gap, i in gaps[:14]:
        print(f"   {gap / 1e3:8.1f}  at {(rows[i][0] - rows[0][0]) / 1e6:6.2f} ms  {rows[i - 1][2].split('(')[0][:48]}  ->  {rows[i][2].split('(')[0][:48]}")


def sequence(path, out):
    """The last fit's kernels in launch order, one line per kernel: start (us from the fit's first kernel), duration, name."""
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    cut = 0
    for i in range(1, len(rows)):
        if rows[i][0] - rows[i - 1][1] > 30_000_000:
            cut = i
    rows = rows[cut:]
    with open(out, "w") as f:
        for s0, e0, n in rows:
            short = n.split("(")[0]
            for junk in ("void at::native::", "void pf::", "at::native::"):
                short = short.replace(junk, "")
            f.write(f"{(s0 - rows[0][0]) / 1e3:10.1f} {(e0 - s0) / 1e3:7.1f}  {short[:110]}\n")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 1000, int(sys.argv[3]) if len(sys.argv) > 3 else 400)
    elif sys.argv[1] == "sequence":
        sequence(sys.argv[2], sys.argv[3])
    else:
        read(sys.argv[2])
