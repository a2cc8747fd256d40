#!/usr/bin/env python
"""Kernel-level micro-benchmark of the fused step: prints the HIP-event average duration of the step kernels for a
matrix of configurations.  Usage: python tools/kbench.py [name ...]   (development tool; bench.py is the contract)."""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _env  # noqa: E402  (tools/_env.py: PF_AMD_LIB / PF_* of this process -> the package's explicit switches)

_env.setup()

from pyfilter_amd import resampling, timeseries as ts  # noqa: E402
from pyfilter_amd.filters.particle import APF, SISR, proposals  # noqa: E402
from pyfilter_amd.timeseries import models  # noqa: E402

dev = "cuda"


def t(v, dtype=torch.float32):
    return torch.tensor(v, dtype=dtype, device=dev)


def make(model, filt, prop, n, b, dtype=torch.float32, resampler="systematic"):
    if model == "sine":
        ssm = ts.LinearStateSpaceModel(models.SineDiffusion(t(0.0, dtype), t(1.0, dtype), dt=0.1), (t(1.0, dtype), t(0.1, dtype)))
        o = ()
    elif model == "lg":
        ssm = ts.LinearStateSpaceModel(models.AR(t(0.0, dtype), t(0.99, dtype), t(0.05, dtype)), (t(1.0, dtype), t(0.15, dtype)))
        o = ()
    elif model == "sv":
        kappa = t([0.05 + 0.01 * (i % 7) for i in range(b)], dtype)
        gamma = t([1.0 + 0.1 * (i % 5) for i in range(b)], dtype)
        sigma = t([0.10 + 0.02 * (i % 3) for i in range(b)], dtype)
        mu = t([0.0 + 0.05 * (i % 4) for i in range(b)], dtype)
        hidden = models.Verhulst(kappa, gamma, sigma, dt=0.2, initial=(t(1.0, dtype), t(0.1, dtype)))
        ssm = models.StochasticVolatilityModel(hidden, mu)
        o = ()
    elif model == "user_sine":  # the README's sine diffusion the reference's way: a python lambda (README.md:44-67)
        from torch.distributions import Normal
        frozen = {}

        def dyn(x, gm, s):
            if os.environ.get("KB_USER_FROZEN"):  # (a callable without launches: what the library's share of a move costs)
                if "f" not in frozen:
                    frozen["f"] = torch.sin(x.value - gm)
                return frozen["f"], s
            return torch.sin(x.value - gm), s

        hidden = ts.AffineEulerMaruyama(dyn, (t(0.0, dtype), t(1.0, dtype)),
                                        Normal(t(0.0, dtype), t(math.sqrt(0.1), dtype)), 0.1, lambda gm, s: Normal(t(0.0, dtype), t(1.0, dtype)))
        if os.environ.get("KB_USER_MEAN"):  # the callable hands over the one-step mean x + f dt itself (no pf_filter_args.user_dt)
            hidden.__class__ = ts.AffineProcess
        hidden.graph_callable = bool(os.environ.get("KB_GRAPH_CALLABLE"))  # the run as one captured hipGraph (callable + library)
        ssm = ts.LinearStateSpaceModel(hidden, (t(1.0, dtype), t(0.1, dtype)))
        o = ()
    elif model == "lorenz":
        hidden = models.Lorenz63(t(10.0, dtype), t(28.0, dtype), t(8.0 / 3.0, dtype), t(1.0, dtype), dt=0.01)
        a = t([[0.8, 0.0, 0.0], [0.0, 0.0, 0.8]], dtype)
        ssm = ts.LinearStateSpaceModel(hidden, (a, t([0.0], dtype), t([math.sqrt(0.1)], dtype)), torch.Size([2]))
        o = (2,)
    cls = {"sisr": SISR, "apf": APF}[filt]
    p = {"bootstrap": proposals.Bootstrap, "lgo": proposals.LinearGaussianObservations}[prop]()
    rs = {"systematic": resampling.systematic, "multinomial": resampling.multinomial}[resampler]
    f = cls(ssm, n, proposal=p, resampling=rs)
    if b > 1:
        f.set_batch_shape(torch.Size([b]))
    return f, o


CONFIGS = {
    "apf_lgo_1m": ("sine", "apf", "lgo", 1 << 20, 1),
    "user_apf_lgo_1m": ("user_sine", "apf", "lgo", 1 << 20, 1),       # lambda-defined model on the fused single-step route
    "user_sisr_boot_1m": ("user_sine", "sisr", "bootstrap", 1 << 20, 1),
    "user_apf_boot_1m": ("user_sine", "apf", "bootstrap", 1 << 20, 1),   # (its first stage needs the callable's plane: a reduce launch per move)
    "user_sisr_lgo_1m": ("user_sine", "sisr", "lgo", 1 << 20, 1),
    "user_apf_lgo_1024x512": ("user_sine", "apf", "lgo", 512, 1024),
    "apf_boot_1m": ("sine", "apf", "bootstrap", 1 << 20, 1),
    "sisr_boot_1m": ("sine", "sisr", "bootstrap", 1 << 20, 1),
    "sisr_boot_lg_1m": ("lg", "sisr", "bootstrap", 1 << 20, 1),
    "apf_lgo_4m": ("sine", "apf", "lgo", 1 << 22, 1),
    "apf_lgo_64x64k": ("sine", "apf", "lgo", 65536, 64),
    "apf_lgo_1024x8k": ("sine", "apf", "lgo", 8192, 1024),
    "apf_sv_64x64k": ("sv", "apf", "bootstrap", 65536, 64),
    "apf_lgo_256k": ("sine", "apf", "lgo", 1 << 18, 1),
    "apf_lgo_512k": ("sine", "apf", "lgo", 1 << 19, 1),
    "apf_lgo_2m": ("sine", "apf", "lgo", 1 << 21, 1),
    "apf_lgo_16x64k": ("sine", "apf", "lgo", 65536, 16),
    "apf_lgo_32x64k": ("sine", "apf", "lgo", 65536, 32),
    # single-tile columns (the reference's own operating point: 1 000 theta x 250-400 particles)
    "apf_lgo_1024x256": ("sine", "apf", "lgo", 256, 1024),
    "apf_lgo_1000x400": ("sine", "apf", "lgo", 400, 1000),
    "apf_lgo_1024x512": ("sine", "apf", "lgo", 512, 1024),
    "sisr_boot_1024x512": ("sine", "sisr", "bootstrap", 512, 1024),
    "apf_sv_1024x512": ("sv", "apf", "bootstrap", 512, 1024),
    "apf_lgo_1024x1024": ("sine", "apf", "lgo", 1024, 1024),
    "apf_lgo_1024x2048": ("sine", "apf", "lgo", 2048, 1024),
    "apf_lgo_1024x4096": ("sine", "apf", "lgo", 4096, 1024),
    "apf_lgo_64x4096": ("sine", "apf", "lgo", 4096, 64),
    "sisr_lorenz_1024x512": ("lorenz", "sisr", "bootstrap", 512, 1024),
    "sisr_lorenz_4m": ("lorenz", "sisr", "bootstrap", 1 << 22, 1),
    "sisr_lorenz_4m_mn": ("lorenz", "sisr", "bootstrap", 1 << 22, 1, "multinomial"),
    "sisr_boot_1m_mn": ("sine", "sisr", "bootstrap", 1 << 20, 1, "multinomial"),
    "apf_lgo_1m_mn": ("sine", "apf", "lgo", 1 << 20, 1, "multinomial"),
    "apf_lgo_4m_mn": ("sine", "apf", "lgo", 1 << 22, 1, "multinomial"),
    "apf_sv_64x64k_mn": ("sv", "apf", "bootstrap", 65536, 64, "multinomial"),
}


def parse_shape(name):
    """``<filter>_<lgo|boot|sv|lorenz>_<B>x<N>`` for shapes without a named entry, e.g. ``apf_lgo_128x512``."""
    filt, kind, shape = name.split("_")[:3]
    b, n = (int(v) for v in shape.split("x"))
    model, prop = {"lgo": ("sine", "lgo"), "boot": ("sine", "bootstrap"), "sv": ("sv", "bootstrap"), "lorenz": ("lorenz", "bootstrap")}[kind]
    return (model, filt, prop, n, b)


def main():
    names = sys.argv[1:] or ["apf_lgo_1m", "apf_boot_1m", "sisr_boot_1m"]
    T = int(os.environ.get("KB_T", 100))
    for name in names:
        cfg = CONFIGS.get(name) or parse_shape(name)
        if cfg[0] == "lorenz" and cfg[4] == 1:  # data simulated from the model (bench.py's generator), not noise
            import bench
            from pyfilter_amd import resampling as rs_
            f, y, w = bench.build_problem("lorenz_mn", torch.float32, torch.device(dev), 1, 0, T)
            f._resampler = {"systematic": rs_.systematic, "multinomial": rs_.multinomial}[cfg[5] if len(cfg) > 5 else "systematic"]
        else:
            f, o = make(*cfg[:5], resampler=(cfg[5] if len(cfg) > 5 else "systematic"))
            g = torch.Generator().manual_seed(0)
            y = (0.3 * torch.randn((T,) + o, generator=g)).cumsum(0).to(dev) if not o else torch.randn((T,) + o, generator=g).to(dev)
        for _ in range(2):  # plan + direct launches, then the hipGraph capture: both outside the timed replays
            f.batch_filter(y, bar=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            res = f.batch_filter(y, bar=False)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps
        k = [0.0, 0.0, 0.0]
        if not os.environ.get("KB_NO_TIMED") and not cfg[0].startswith("user"):  # (the replays of the timed run would pollute a PMC profile)
            f._time_kernels = True
            f.batch_filter(y, bar=False)
            k = f.kernel_ms
        n, b = cfg[3], cfg[4]
        if int(os.environ.get("PF_DEBUG_CUT", "0")) < 0:
            import ctypes as C
            from pyfilter_amd import _lib as L
            off = C.c_size_t(0)
            lib = L.load()
            lib.pf_debug_offset.argtypes = [C.c_int64, C.c_int64, C.POINTER(C.c_size_t)]
            lib.pf_debug_offset(cfg[3], cfg[4], C.byref(off))
            ws = f._last_run["ws"]
            st = ws[off.value:off.value + 256].view(torch.int64).cpu().tolist()
            sp = [st[i] - st[0] for i in (1, 2, 8, 9, 10, 11, 12, 13, 14, 15)]
            t0w = min(st[16:24])
            print("   wall clock (10 ns ticks since the first of 8 sampled workgroups, tiles 0,128,..,896): start", [v - t0w for v in st[16:24]],
                  " end", [v - t0w for v in st[24:32]])
            print("   stamps of workgroup", -int(os.environ["PF_DEBUG_CUT"]) - 1, "(clock64 ticks from kernel entry: table built, window start known | body entry, "
                  "pre-loop, loads issued + normals, ancestors, compute + stores, push, pre-finish, end):", sp)
        print(f"{name:22s} us/step {1e6 * wall / T:8.2f}  particle-steps/s {n * b * T / wall:10.3e}  kernels(us, event-bracketed) "
              f"scan {1e3 * k[1]:7.2f} step {1e3 * k[2]:7.2f}  ll {res.loglikelihood.reshape(-1)[0].item():.3f}", flush=True)


if __name__ == "__main__":
    main()
