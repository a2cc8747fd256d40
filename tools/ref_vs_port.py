"""DEVELOPMENT TOOL (runs only in the container that holds /root/reference): times the UNMODIFIED reference
(imported behind oracle/ref_shim) against the port bench.py times as `cpu_baseline` (oracle/cpu_ref.py), same model,
same N, same thread count, fp32 - the evidence behind "the port is not slower than the reference".

    python tools/ref_vs_port.py [--N 1048576] [--steps 6] [--threads 8]   > profiles/r03_ref_vs_port.txt
"""
import argparse
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle", "ref_shim"))
sys.path.insert(1, "/root/reference")
sys.path.insert(2, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=1 << 20)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--threads", type=int, default=8)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    torch.manual_seed(0)

    from pyfilter.filters.particle import APF, SISR, proposals  # the reference
    from stochproc import timeseries as ts
    from torch.distributions import Normal

    from oracle import cpu_ref, models as M

    n, steps = args.N, args.steps
    y = torch.randn(steps + 1)
    rows = []
    for name, filt_cls, fname, pname in (("APF + LinearGaussianObservations (BASELINE configs[1])", APF, "apf", "lgo"),
                                         ("SISR + Bootstrap", SISR, "sisr", "bootstrap")):
        # the sine diffusion of README.md:44-67 at bench.py's parameters
        inc = Normal(torch.tensor(0.0), torch.tensor(math.sqrt(0.1)))
        hidden = ts.AffineEulerMaruyama(lambda x, g, s: (torch.sin(x.value - g), s), (0.0, 1.0), inc, 0.1,
                                        lambda *_: Normal(torch.tensor(0.0), torch.tensor(1.0)))
        ssm = ts.LinearStateSpaceModel(hidden, (1.0, 0.1), torch.Size([]))
        prop = {"lgo": proposals.LinearGaussianObservations, "bootstrap": proposals.Bootstrap}[pname]()
        ref = filt_cls(ssm, n, proposal=prop)
        ref.batch_filter(y[:1], bar=False)  # warm-up
        t_ref = []
        for _ in range(3):  # whole batch_filter calls of `steps` observations: what bench.py's cpu_baseline leg times
            t0 = time.perf_counter()
            res = ref.batch_filter(y[:steps], bar=False)
            _ = res.loglikelihood.item()
            t_ref.append((time.perf_counter() - t0) / steps)

        spec = M.ModelSpec(M.HID_SINE_EM, (0.0, 1.0), 0, 0.1, (0.0, 1.0), M.OBS_LINEAR, (1.0, 0.0, 0.1), 0)
        x0 = torch.randn(n)
        cpu_ref.batch_filter(spec, fname, pname, y[:1], x0, None, None)
        t_port = []
        for _ in range(3):
            t0 = time.perf_counter()
            cpu_ref.batch_filter(spec, fname, pname, y[:steps], x0, None, None)
            t_port.append((time.perf_counter() - t0) / steps)
        med = lambda v: sorted(v)[len(v) // 2]  # noqa: E731
        rows.append((name, med(t_ref), min(t_ref), med(t_port), min(t_port)))

    print(f"reference (imported, unmodified) vs port (oracle/cpu_ref.py): N = {n}, fp32, {args.threads} threads, torch {torch.__version__}, "
          f"3 batch_filter calls of {steps} observations each (ms per time step), {os.cpu_count()} logical CPUs")
    print(f"{'configuration':60s} {'ref median ms':>14s} {'ref min ms':>11s} {'port median ms':>15s} {'port min ms':>12s} {'port / ref':>10s}")
    for name, rm, rmin, pm, pmin in rows:
        print(f"{name:60s} {1e3 * rm:14.1f} {1e3 * rmin:11.1f} {1e3 * pm:15.1f} {1e3 * pmin:12.1f} {pm / rm:10.2f}")
    print("port / ref < 1: the timed CPU baseline is FASTER than the reference itself would be (the GPU / CPU ratio is conservative)")


if __name__ == "__main__":
    main()
