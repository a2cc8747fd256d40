#!/usr/bin/env python
"""The oracle's float64 answer for the data ``bench.py`` times (development tool, CPU only): the log-likelihood of the seeded
observations of a workload (``bench.build_problem``: generator seed 123 + rank, rank 0) under the workload's model, from
``oracle/cpu_ref.py`` (the torch-CPU restatement of the reference's filters) in float64 at the workload's particle count, for
two independent draw seeds.  ``bench.py`` carries the value (``EXPECTED_LL``) and checks the log-likelihood of its own timed
passes against it.  Usage: python tools/bench_reference_ll.py [workload] [N] [seeds]  ->  profiles/r04_bench_reference_ll.txt"""
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def data(name, t_len):
    """bench.build_problem's observations of rank 0, drawn exactly as it draws them (no GPU, no product import)."""
    gen = torch.Generator().manual_seed(123)
    if name == "apf_lgo_1m":
        x, ys = torch.randn((), generator=gen).item(), []
        for _ in range(t_len):
            x = x + math.sin(x) * 0.1 + math.sqrt(0.1) * torch.randn((), generator=gen).item()
            ys.append(x + 0.1 * torch.randn((), generator=gen).item())
        return torch.tensor(ys, dtype=torch.float32).double()  # (bench hands the filter float32 observations)
    raise KeyError(name)


def main():
    from oracle import cpu_ref, models as M

    name = sys.argv[1] if len(sys.argv) > 1 else "apf_lgo_1m"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
    seeds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    t_len = 250
    y = data(name, t_len)
    spec = M.ModelSpec(M.HID_SINE_EM, (0.0, 1.0), 0, 0.1, (0.0, 1.0), M.OBS_LINEAR, (1.0, 0.0, 0.1), 0)
    torch.set_default_dtype(torch.float64)
    for s in range(seeds):
        torch.manual_seed(1000 + s)
        t0 = time.perf_counter()
        x0 = M.initial_sample(spec, torch.randn(n, 1, dtype=torch.float64))
        z0 = None
        r = cpu_ref.batch_filter(spec, "apf", "lgo", y, x0, z0, torch.rand(t_len, 1, dtype=torch.float64))
        print(f"{name} N={n} T={t_len} float64 oracle, draw seed {1000 + s}: loglikelihood = {r['loglikelihood'].item():.6f} "
              f"last filter mean = {r['filter_means'][-1].reshape(-1)[0].item():.6f}  ({time.perf_counter() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
