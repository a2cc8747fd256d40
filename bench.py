#!/usr/bin/env python
"""
bench.py - throughput of the SISR/APF hot path on MI355X, BASELINE.json's metric:

    particle-steps/s = batch x particles x T / wall-time   (+ achieved HBM GB/s of the dominant kernel)

A "step" (``--steps K``) is one full ``batch_filter`` pass of the fused HIP loop over T synthetic observations with all
inputs already resident in HBM.  Workloads (``--workload``):

    apf_lgo_1m   (default) BASELINE.json configs[1]: sine diffusion, APF + LinearGaussianObservations,
                 1 048 576 particles, T = 250, systematic resampling
    sv_batch     configs[2]: Verhulst SV, APF + Bootstrap, 64 series x 65 536 particles
    lorenz_mn    configs[3]: Lorenz-63, SISR + Bootstrap, 4 194 304 particles, multinomial resampling
    smc2_shard   configs[4]: one theta-shard of SMC^2 (1 024 / n_gpus filters x 8 192 particles, T = 500), the filtering
                 pass alone (one batch_filter call per step of the bench, log-likelihoods all-gathered at its end)
    smc2         configs[4] as the algorithm: ``pyfilter_amd.inference.SMC2`` on the OU model / priors of the reference's
                 tests/inference/models.py - 1 024 theta-particles block-sharded over the ranks, one fused filter() move
                 per observation, the theta-weights all-gathered EVERY observation (RCCL), ESS-triggered PMMH
                 rejuvenations with filter-state redistribution.  Counts the T filtering moves of all theta-particles;
                 rejuvenation work is overhead inside the timed region.

Multi-GPU: one process per GPU over RCCL.  ``python bench.py --gpus N`` on its own spawns its N ranks (re-executes itself
under ``python -m torch.distributed.run --nproc-per-node N``); launched under torchrun it reads RANK / WORLD_SIZE from the
environment.  What shards is the filters' batch dimension - SMC^2's theta-particles - so with N > 1 the default workload is
``smc2`` (BASELINE configs[4]: 1 024 theta-particles block-sharded over the ranks, STRONG scaling, the theta-weights
all-gathered per block of observations, whole filters redistributed by all-to-all on a rejuvenation); its JSON line also
carries the same job timed on ONE of the GPUs (``single_gpu_same_workload``) so that the strong-scaling ratio can be read
off one line.  A single filter (configs[1], [3]) does not shard - that would need a cross-GPU scan - so
``--workload apf_lgo_1m`` etc. with N > 1 run one replica per GPU (weak scaling) and all-gather the log-likelihoods.

Prints ONE JSON line on rank 0 (contract in the task statement) incl. ``roofline`` and ``cpu_baseline``.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); a plain device copy measures 5.1-6.3 TB/s (roofline.measured_copy_ceiling)

WORKLOADS = {
    # name: (filter, proposal, resampler, N, B_total, D, T)
    "apf_lgo_1m": dict(filter="apf", proposal="lgo", resampler="systematic", N=1 << 20, B=1, D=1, T=250),
    "sv_batch": dict(filter="apf", proposal="bootstrap", resampler="systematic", N=65536, B=64, D=1, T=1000),
    "lorenz_mn": dict(filter="sisr", proposal="bootstrap", resampler="multinomial", N=1 << 22, B=1, D=3, T=2000),
    "smc2_shard": dict(filter="apf", proposal="bootstrap", resampler="systematic", N=8192, B=1024, D=1, T=500),
    "smc2": dict(filter="apf", proposal="lgo", resampler="systematic", N=8192, B=1024, D=1, T=500),
}

# What the timed passes must compute: the log-likelihood of rank 0's seeded observations (build_problem: generator seed 123)
# under the workload's model, from the ORACLE (oracle/cpu_ref.py, float64, CPU, the workload's own N and T) - two independent
# draw seeds of tools/bench_reference_ll.py (profiles/r04_bench_reference_ll.txt): -88.282269 and -88.299229.  `tol`: Monte-Carlo
# spread of a 2^20-particle filter over 250 steps (0.02) + the float32 path's distance from exact arithmetic (BASELINE.md
# section 2), with room.  bench.py refuses to print a throughput for a pass that computed something else.
EXPECTED_LL = {"apf_lgo_1m": {"loglikelihood": -88.2907, "tol": 0.15, "N": 1 << 20, "T": 250,
                              "source": "oracle/cpu_ref.py float64, tools/bench_reference_ll.py"}}


def check_loglikelihood(workload, w, ll_all):
    """Every timed pass's output is checked, not just printed: finite for every filter of the job, and - for a workload with
    a committed oracle value at its default size - within the stored tolerance of it."""
    ll = ll_all.reshape(-1).double().cpu()
    out = {"finite": bool(torch.isfinite(ll).all()), "sample": float(ll[0])}
    if not out["finite"]:
        raise AssertionError(f"bench: non-finite log-likelihood in the timed pass of {workload}: {ll[:8].tolist()}")
    exp = EXPECTED_LL.get(workload)
    if exp is not None and w["N"] == exp["N"] and w["T"] == exp["T"]:
        out.update(expected=exp["loglikelihood"], tol=exp["tol"], abs_diff=abs(out["sample"] - exp["loglikelihood"]), source=exp["source"])
        if out["abs_diff"] > exp["tol"]:
            raise AssertionError(f"bench: the timed pass of {workload} computed loglikelihood {out['sample']:.4f}, the oracle's "
                                 f"float64 value for this data is {exp['loglikelihood']:.4f} (tolerance {exp['tol']})")
    return out


def run_smc2(w, dtype, device, world, rank, steps, warmup, t_override=None, solo=False):
    """BASELINE configs[4] as the algorithm; returns (elapsed seconds for `steps` full fits, info).  ``solo``: the whole
    job (``w["B"]`` theta-particles) on this rank alone, no sharding, no collectives (the single-GPU reference leg of a
    multi-GPU line).  ``w["B"]`` is the TOTAL number of theta-particles: 1 024 (strong scaling) or 1 024 per GPU (weak)."""
    import torch.distributed as dist
    from torch.distributions import Exponential, LogNormal, Normal

    from pyfilter_amd import timeseries as ts
    from pyfilter_amd.filters.particle import APF, proposals
    from pyfilter_amd.inference import SMC2
    from pyfilter_amd.timeseries import models

    t_len = t_override or w["T"]
    g = torch.Generator().manual_seed(123)  # tests/inference/models.py:13-19: OU(0.025, 0, 0.05), y = x + 0.05 v
    x, ys = 0.0, []
    for _ in range(t_len):
        x = x * math.exp(-0.025) + 0.05 * math.sqrt((1 - math.exp(-0.05)) / 0.05) * torch.randn((), generator=g).item()
        ys.append(x + 0.05 * torch.randn((), generator=g).item())
    y = torch.tensor(ys, dtype=dtype, device=device)
    priors = {"kappa": Exponential(10.0), "gamma": Normal(0.0, 1.0), "sigma": LogNormal(-2.0, 1.0)}  # models.py:29-31

    # the observation constants live on the device ONCE: created inside the builder they would be two pageable host -> device
    # copies - two waits for the device - per model build, and PMMH rebuilds the model at every move
    obs_a, obs_s = torch.tensor(1.0, dtype=dtype, device=device), torch.tensor(0.05, dtype=dtype, device=device)

    def build(theta):
        return ts.LinearStateSpaceModel(models.OrnsteinUhlenbeck(theta["kappa"], theta["gamma"], theta["sigma"], dt=1.0), (obs_a, obs_s))

    def fit(seed):
        filt = APF(build, w["N"], proposal=proposals.LinearGaussianObservations(), seed=2024 + seed)
        from pyfilter_amd.distributed import SOLO

        alg = SMC2(filt, w["B"], priors, threshold=0.2, device=device, dtype=dtype, seed=seed, group=SOLO if solo else None)
        state = alg.fit(y)
        return alg, state

    def barrier():
        torch.cuda.synchronize()
        if world > 1 and not solo:
            dist.barrier()
            torch.cuda.synchronize()

    for k in range(warmup):
        fit(k)
    barrier()
    t0 = time.perf_counter()
    for k in range(steps):
        alg, state = fit(100 + k)
    barrier()
    elapsed = time.perf_counter() - t0
    info = {"rejuvenations": len(alg._kernel.acceptance_history), "particle_increases": alg._kernel._increases,
            "state_particles_at_end": int(alg.filter.particles[0]), "posterior_mean": alg.posterior_mean(state).tolist(),
            "theta_per_rank": alg.shard.local, "T": t_len, "block": alg._block}
    return elapsed, info, state.global_weights()



def _rccl_version(world):
    """The RCCL the collectives ran on - None for one rank, and None when the ranks talk over gloo (a development run of several
    ranks on ONE GPU, PF_BENCH_SHARE_GPU=1: no RCCL involved)."""
    if world <= 1:
        return None
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_backend() != "nccl":
        return None
    return ".".join(str(v) for v in torch.cuda.nccl.version())


def smc2_step_kernel_roofline(w, b_local, dtype, device, t_len=64):
    """The dominant kernel of the SMC^2 job - ``k_fused_step`` at the PER-RANK shape (``b_local`` filters x 8 192 particles,
    APF + optimal proposal on the OU model: what every online move and every PMMH re-filter launches) - timed with HIP events
    on the launch stream (``pf_filter_run_timed``) on a filter of exactly that shape, priced on SURVEY 8(d)'s APF bytes."""
    from pyfilter_amd import timeseries as ts
    from pyfilter_amd.filters.particle import APF, proposals
    from pyfilter_amd.timeseries import models

    gen = torch.Generator().manual_seed(77)
    t = lambda v: torch.tensor(v, dtype=dtype, device=device)  # noqa: E731
    kappa = (0.01 + 0.05 * torch.rand(b_local, generator=gen)).to(dtype).to(device)
    gamma = (0.2 * torch.randn(b_local, generator=gen)).to(dtype).to(device)
    sigma = (0.03 + 0.04 * torch.rand(b_local, generator=gen)).to(dtype).to(device)
    ssm = ts.LinearStateSpaceModel(models.OrnsteinUhlenbeck(kappa, gamma, sigma, dt=1.0), (t(1.0), t(0.05)))
    filt = APF(ssm, w["N"], proposal=proposals.LinearGaussianObservations(), seed=9)
    filt.set_batch_shape(torch.Size([b_local]))
    y = (0.1 * torch.randn(t_len, generator=gen)).to(dtype).to(device)
    filt.batch_filter(y, bar=False)
    filt._time_kernels = True
    filt.batch_filter(y, bar=False)
    torch.cuda.synchronize()
    filt._time_kernels = False
    step_ms = filt.kernel_ms[2]
    from pyfilter_amd import ops

    spec = ops.debug_launch_trace(1)[-1]["SPEC"]  # which route the library took at this shape: 10 = the column-cluster kernel
    kname = {10: "k_fused_cluster", 9: "k_fused_column"}.get(spec, "k_fused_step")
    esz = 8 if dtype == torch.float64 else 4
    bm = byte_models(dict(w, D=1, filter="apf"), esz)
    units = w["N"] * b_local
    gbs = {k: v * units / (step_ms * 1e-3) / 1e9 for k, v in bm.items()}
    return {
        "bound": "valu" if gbs["as_built"] / HBM_PEAK_GBS < 0.30 else "hbm", "priced_against": "hbm",
        "kernel": kname, "achieved": gbs["survey_8d"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": gbs["survey_8d"] / HBM_PEAK_GBS, "traffic": None,
        "route": {"k_fused_cluster": "column-cluster: ceil(N / 1024) workgroups per filter hold it in registers for the whole run "
                                     "(one launch per block of moves; the state's HBM traffic is its exchange through L2)",
                  "k_fused_column": "column-persistent", "k_fused_step": "one launch per time step"}[kname],
        "shape": f"{b_local} filters x {w['N']} particles per launch (this rank's theta-block), APF + lgo, OU",
        "byte_model": "survey_8d: SURVEY.md 8(d) APF bytes (32 + 16 D) x particles per launch / the step kernel's in-sequence duration",
        "bytes_per_particle": bm, "bytes_per_launch": {k: v * units for k, v in bm.items()},
        "kernel_us": {kname: 1e3 * step_ms},
        "duration_source": "HIP events on the launch stream around a 64-move run at this shape / 64 (pf_filter_run_timed)",
        "as_built": {"achieved": gbs["as_built"], "frac": gbs["as_built"] / HBM_PEAK_GBS},
    }


def smc2_cpu_baseline(w, seconds_budget=12.0, b_sample=8):
    """The oracle on the host cores for the FILTERING moves of the SMC^2 job (what ``value`` counts): APF + optimal proposal on
    the OU model, ``b_sample`` theta-particles x 8 192 state particles, as many observations as fit the budget - the
    rejuvenations (re-filtering all parsed data per PMMH move) would come on top, so this flatters the CPU."""
    from oracle import cpu_ref
    from oracle import models as M

    g = torch.Generator().manual_seed(1)
    n, b = w["N"], b_sample
    kappa, gamma = 0.01 + 0.05 * torch.rand(b, generator=g), 0.2 * torch.randn(b, generator=g)
    sigma = 0.03 + 0.04 * torch.rand(b, generator=g)
    spec = M.ModelSpec(M.HID_OU, (kappa, gamma, sigma), 0, 1.0, (0.0, 0.1), M.OBS_LINEAR, (1.0, 0.0, 0.05), 0)
    y, x0 = 0.1 * torch.randn(4096, generator=g), 0.1 * torch.randn(n, b, generator=g)

    def run(steps):
        t0 = time.perf_counter()
        cpu_ref.batch_filter(spec, "apf", "lgo", y[:steps], x0, None, None)
        return time.perf_counter() - t0

    cores = os.cpu_count() or 1
    best = None
    for th in sorted({c for c in (1, 8, 16, 32) if c <= cores}):
        torch.set_num_threads(th)
        run(1)
        dt = min(run(2) for _ in range(2)) / 2
        if best is None or dt < best[1]:
            best = (th, dt)
    torch.set_num_threads(best[0])
    steps = int(max(3, min(400, seconds_budget / best[1])))
    dt = run(steps)
    return {"value": n * b * steps / dt, "unit": "particle-steps/s", "cores": best[0], "kind": "port",
            "sample": f"smc2: the filtering moves only - APF + lgo on the OU model, {b} theta-particles x {n} state particles, {steps} "
                      f"observations ({dt:.1f} s) of T={w['T']}, fp32, oracle/cpu_ref.py on {best[0]} threads (best of 1/8/16/32); no "
                      f"rejuvenation work (the GPU figure includes it as overhead)",
            "host": host_info()}


def smc2_line(args, dtype, device, world, rank, scaling, attach=False):
    """One JSON-able record of the SMC^2 job (BASELINE configs[4]).  ``scaling``: ``"strong"`` - 1 024 theta-particles split over
    the ranks; ``"weak"`` - 1 024 per GPU.  At N = 1 both are the same job: the N = 1 point of either curve."""
    import torch.distributed as dist

    w = dict(WORKLOADS["smc2"])
    per_gpu = w["B"]
    if scaling == "weak":
        w["B"] = per_gpu * world
    elapsed, info, _ = run_smc2(w, dtype, device, world, rank, args.steps, args.warmup, args.T)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = tmax.item()
    solo = None
    if world > 1:  # the one-GPU job of the same curve on ONE of these GPUs (rank 0 alone, the others wait)
        if rank == 0:
            try:  # (rank 0 alone: whatever happens here it must reach the barrier the others wait at)
                k = max(1, min(2, args.steps))
                w1 = dict(w, B=per_gpu)
                e1, _, _ = run_smc2(w1, dtype, device, world, rank, k, 1, args.T, solo=True)
                solo = {"value": w1["N"] * w1["B"] * info["T"] * k / e1, "unit": "particle-steps/s", "ms_per_step": 1e3 * e1 / k,
                        "what": f"{per_gpu} theta-particles on rank 0's GPU alone (no sharding, no collectives), timed after the "
                                f"N-GPU region: the N = 1 point of this curve"}
            except Exception as exc:
                solo = {"error": f"{type(exc).__name__}: {exc}"}
        dist.barrier()
    roof = None
    if rank == 0:
        try:
            roof = smc2_step_kernel_roofline(w, info["theta_per_rank"], dtype, device)
        except Exception as exc:
            roof = {"error": f"{type(exc).__name__}: {exc}"}
    if world > 1:
        dist.barrier()
    if rank != 0:
        return None
    value = w["N"] * w["B"] * info["T"] * args.steps / elapsed
    if solo and "value" in solo:
        solo["speedup_over_it"] = value / solo["value"]
    cpu = None
    if world == 1 and not args.no_cpu_baseline and not attach:
        cpu = smc2_cpu_baseline(dict(w, T=info["T"]))
    out = {
        "metric": f"particle-steps/sec (batch x particles x T), SMC^2 {w['B']} theta x {w['N']} particles", "value": value,
        "unit": "particle-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic", "world_size": world,
        "rccl_version": _rccl_version(world),
        "config": {"workload": f"smc2: SMC^2, APF + lgo, {w['B']} theta-particles (sharded {info['theta_per_rank']} per GPU) x "
                               f"{w['N']} state particles, T={info['T']}, theta-weights all-gathered per block of {info.get('block', 16)} observations",
                   "parallelism": f"theta-particles block-sharded over {world} GPU(s)", **info},
        "single_gpu_same_workload": solo, "roofline": roof, "cpu_baseline": cpu}
    if cpu:
        out["speedup_vs_cpu_baseline"] = value / cpu["value"]
    return out


def build_problem(name, dtype, device, world, rank, t_override=None, n_override=None):
    from pyfilter_amd import resampling, timeseries as ts
    from pyfilter_amd.filters.particle import APF, SISR, proposals
    from pyfilter_amd.timeseries import models

    w = dict(WORKLOADS[name])
    if t_override:
        w["T"] = t_override
    if n_override:
        w["N"] = n_override
    t = lambda v: torch.tensor(v, dtype=dtype, device=device)  # noqa: E731
    gen = torch.Generator().manual_seed(123 + rank)
    b = w["B"]
    if name == "apf_lgo_1m":
        hidden = models.SineDiffusion(t(0.0), t(1.0), dt=0.1)
        ssm = ts.LinearStateSpaceModel(hidden, (t(1.0), t(0.1)))
        x, ys = torch.randn((), generator=gen).item(), []
        for _ in range(w["T"]):
            x = x + math.sin(x) * 0.1 + math.sqrt(0.1) * torch.randn((), generator=gen).item()
            ys.append(x + 0.1 * torch.randn((), generator=gen).item())
        y = torch.tensor(ys, dtype=dtype)
    elif name == "sv_batch":
        kappa = 0.05 + 0.01 * torch.rand(b, generator=gen)
        gamma = 1.0 + 0.2 * torch.rand(b, generator=gen)
        sigma = 0.10 + 0.05 * torch.rand(b, generator=gen)
        mu = 0.05 * torch.randn(b, generator=gen)
        hidden = models.Verhulst(kappa.to(dtype).to(device), gamma.to(dtype).to(device), sigma.to(dtype).to(device),
                                 dt=0.2, initial=(t(1.0), t(0.1)))
        ssm = models.StochasticVolatilityModel(hidden, mu.to(dtype).to(device))
        v, ys = torch.ones(b), []
        for _ in range(w["T"]):
            v = (v + kappa * (gamma - v) * v * 0.2 + sigma * v * math.sqrt(0.2) * torch.randn(b, generator=gen)).clamp_min(1e-3)
            ys.append(mu + v * torch.randn(b, generator=gen))
        y = torch.stack(ys).to(dtype)
    elif name == "lorenz_mn":
        hidden = models.Lorenz63(t(10.0), t(28.0), t(8.0 / 3.0), t(1.0), dt=0.01)
        a = t([[0.8, 0.0, 0.0], [0.0, 0.0, 0.8]])
        ssm = ts.LinearStateSpaceModel(hidden, (a, t([0.0]), t([math.sqrt(0.1)])), torch.Size([2]))
        x = torch.tensor([-5.91652, -5.52332, 24.5723], dtype=torch.float64)
        ys = []
        for _ in range(w["T"]):
            f = torch.stack((-10.0 * (x[0] - x[1]), 28.0 * x[0] - x[1] - x[0] * x[2], x[0] * x[1] - 8.0 / 3.0 * x[2]))
            x = x + f * 0.01 + 0.1 * torch.randn(3, generator=gen, dtype=torch.float64)
            ys.append(0.8 * x[[0, 2]] + math.sqrt(0.1) * torch.randn(2, generator=gen, dtype=torch.float64))
        y = torch.stack(ys).to(dtype)
    elif name == "smc2_shard":
        b = w["B"] = max(1, w["B"] // world)  # theta-particles block-sharded across the ranks
        kappa = 0.01 + 0.05 * torch.rand(b, generator=gen)
        gamma = 0.2 * torch.randn(b, generator=gen)
        sigma = 0.03 + 0.04 * torch.rand(b, generator=gen)
        hidden = models.OrnsteinUhlenbeck(kappa.to(dtype).to(device), gamma.to(dtype).to(device),
                                          sigma.to(dtype).to(device), dt=1.0)
        ssm = ts.LinearStateSpaceModel(hidden, (t(1.0), t(0.05)))
        g2 = torch.Generator().manual_seed(123)  # the data are shared by all shards
        x, ys = 0.0, []
        for _ in range(w["T"]):
            x = x * math.exp(-0.025) + 0.05 * math.sqrt((1 - math.exp(-0.05)) / 0.05) * torch.randn((), generator=g2).item()
            ys.append(x + 0.05 * torch.randn((), generator=g2).item())
        y = torch.tensor(ys, dtype=dtype)
    else:
        raise KeyError(name)

    cls = {"sisr": SISR, "apf": APF}[w["filter"]]
    prop = {"bootstrap": proposals.Bootstrap, "lgo": proposals.LinearGaussianObservations}[w["proposal"]]()
    rs = {"systematic": resampling.systematic, "multinomial": resampling.multinomial}[w["resampler"]]
    filt = cls(ssm, w["N"], proposal=prop, resampling=rs, seed=2024 + rank)
    if w["B"] > 1 or name in ("sv_batch", "smc2_shard"):
        filt.set_batch_shape(torch.Size([w["B"]]))
    return filt, y.to(device), w


def byte_models(w, e=4):
    """HBM bytes per particle per time step, two accountings (e = bytes per state / weight element):

    * ``survey_8d`` - SURVEY.md section 8(d)'s contract figure: the compulsory traffic of the reference's dataflow (a reduce
      pass, a scan pass and a fused search / gather / propagate / weight pass; int64 ancestors): 32 + 12 D for SISR,
      32 + 16 D for APF (fp32);
    * ``as_built`` - what the one kernel of a step has to move here: it reads the local scans L (e) and x[anc] (e D),
      writes x' (e D), logw' (e), the next L (e), anc (int32) = 16 + 8 D (fp32) - the separate reduce and scan passes no
      longer exist.  ``roofline.traffic`` (PMC) is to be read against this one."""
    d = w["D"]
    k = e // 4
    survey = k * (32 + (16 if w["filter"] == "apf" else 12) * d)
    return {"survey_8d": survey, "as_built": e * (3 + 2 * d) + 4}


def device_copy_ceiling(device, n=1 << 28, reps=5):
    """What a plain copy kernel reaches on this GPU (GB/s, read + written bytes): the practical HBM ceiling next to the
    vendor figure."""
    a = torch.empty(n, dtype=torch.float32, device=device).fill_(1.0)
    b = torch.empty_like(a)
    for _ in range(2):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2 * 4 * n * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def host_info():
    """CPU model, socket / core counts and a measured memory-copy ceiling of this box."""
    import subprocess

    info = {"logical_cpus": os.cpu_count()}
    try:
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        kv = {ln.split(":", 1)[0].strip(): ln.split(":", 1)[1].strip() for ln in txt.splitlines() if ":" in ln}
        info["model"] = kv.get("Model name")
        sockets, cps = int(kv.get("Socket(s)", 1)), int(kv.get("Core(s) per socket", 0))
        info["physical_cores"] = sockets * cps if cps else None
        info["threads_per_core"] = int(kv.get("Thread(s) per core", 1))
    except Exception:
        pass
    try:  # copy ceiling: out-of-place copy of 1 GiB of float32 with torch's intra-op threads (read + write bytes / time)
        src = torch.empty(1 << 28, dtype=torch.float32).fill_(1.0)
        dst = torch.empty_like(src).fill_(0.0)  # (pages touched before the timed copies)
        best = None
        for _ in range(6):
            t0 = time.perf_counter()
            dst.copy_(src)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        info["copy_GBs"] = 2 * src.numel() * 4 / best / 1e9
        info["copy_threads"] = torch.get_num_threads()
    except Exception:
        pass
    return info


def oracle_problem(name, w):
    """The oracle-side description of a workload: (ModelSpec, y, x0) with the parameters of ``build_problem``."""
    from oracle import models as M

    g = torch.Generator().manual_seed(1)
    n, b = w["N"], w["B"]
    if name == "apf_lgo_1m":
        spec = M.ModelSpec(M.HID_SINE_EM, (0.0, 1.0), 0, 0.1, (0.0, 1.0), M.OBS_LINEAR, (1.0, 0.0, 0.1), 0)
        return spec, torch.randn(4096, generator=g), torch.randn(n)  # unbatched, the layout the reference is fastest in
    if name == "sv_batch":
        kappa, gamma = 0.05 + 0.01 * torch.rand(b, generator=g), 1.0 + 0.2 * torch.rand(b, generator=g)
        sigma, mu = 0.10 + 0.05 * torch.rand(b, generator=g), 0.05 * torch.randn(b, generator=g)
        spec = M.ModelSpec(M.HID_VERHULST_EM, (kappa, gamma, sigma), 0, 0.2, (1.0, 0.1), M.OBS_SV, (mu,), 0)
        return spec, 0.05 + torch.randn(4096, b, generator=g), 1.0 + 0.1 * torch.randn(n, b, generator=g)
    if name == "lorenz_mn":
        a = torch.tensor([[0.8, 0.0, 0.0], [0.0, 0.0, 0.8]])
        m0, s0 = torch.tensor([-5.91652, -5.52332, 24.5723]), torch.full((3,), math.sqrt(10.0))
        spec = M.ModelSpec(M.HID_LORENZ63_EM, (10.0, 28.0, 8.0 / 3.0, 1.0), 3, 0.01, (m0, s0), M.OBS_LINEAR,
                           (a, torch.tensor([0.0]), torch.tensor([math.sqrt(0.1)])), 2)
        y = torch.tensor([-4.7, 19.6]) + 0.3 * torch.randn(4096, 2, generator=g)
        return spec, y, m0 + 0.3 * torch.randn(n, 3, generator=g)
    if name == "smc2_shard":
        kappa, gamma = 0.01 + 0.05 * torch.rand(b, generator=g), 0.2 * torch.randn(b, generator=g)
        sigma = 0.03 + 0.04 * torch.rand(b, generator=g)
        spec = M.ModelSpec(M.HID_OU, (kappa, gamma, sigma), 0, 1.0, (0.0, 0.1), M.OBS_LINEAR, (1.0, 0.0, 0.05), 0)
        return spec, 0.1 * torch.randn(4096, generator=g), 0.1 * torch.randn(n, b, generator=g)
    raise KeyError(name)


def cpu_baseline(name, w, seconds_budget=12.0):
    """The oracle (torch-CPU restatement of the reference's aten-op sequence, ``kind: "port"``) timed on this box's host
    cores on a bounded sample of the same workload: same N, B, model, filter, proposal, resampler, fp32; as many time
    steps as fit the budget (configs[3], ~2 s per step, is extrapolated from a handful of steps as BASELINE.md section 3
    says).  Reported: the best thread count of a short sweep AND the single-thread figure, the CPU model, the physical
    core count and a measured copy ceiling."""
    from oracle import cpu_ref

    spec, y, x0 = oracle_problem(name, w)
    cores = os.cpu_count() or 1
    n, b = w["N"], w["B"]

    def run(steps):
        t0 = time.perf_counter()
        cpu_ref.batch_filter(spec, w["filter"], w["proposal"], y[:steps], x0, None, None, resampler=w["resampler"])
        return time.perf_counter() - t0

    host = host_info()
    # thread count: torch oversubscribes badly on many-core hosts (256 threads were 100x slower than 16 on the GPU
    # box), so give the CPU path its best case: a short sweep, keep the fastest
    best, sweep = None, {}
    for th in sorted({c for c in (8, 16, 32, 64, 128) if c <= cores} | {min(cores, 8)}):
        torch.set_num_threads(th)
        first = run(1)  # (warm-up of this thread count; also tells how many timed steps the sweep can afford)
        reps = 5 if first < 0.5 else 2
        times = sorted(run(1) for _ in range(reps))
        dt_min, dt_med = times[0], times[len(times) // 2]
        sweep[th] = {"min": n * b / times[-1], "median": n * b / dt_med, "max": n * b / dt_min, "steps_timed": reps}
        if best is None or dt_med < best[1]:
            best = (th, dt_med)
        if dt_med > 3.0 * best[1]:
            break
    torch.set_num_threads(1)
    run(1)
    k1 = int(max(2, min(8, 3.0 / max(1e-3, best[1] * best[0] / 2))))
    dt1 = run(k1)
    one_thread = n * b * k1 / dt1
    cores_used = best[0]
    torch.set_num_threads(cores_used)
    steps = int(max(3, min(400, seconds_budget / best[1])))
    dt = run(steps)
    return {
        "value": n * b * steps / dt, "unit": "particle-steps/s", "cores": cores_used, "kind": "port",
        "sample": f"{name}: N={n}, B={b}, {steps} time steps ({dt:.1f} s) of T={w['T']}, fp32, torch {torch.__version__} CPU, "
                  f"{cores_used} threads (best median of a 8..128 sweep, {sweep[cores_used]['steps_timed']} one-step timings per count, "
                  f"on {os.cpu_count()} logical CPUs); oracle/cpu_ref.py (same aten-op sequence as the reference; timed against "
                  f"the imported reference by tools/ref_vs_port.py -> profiles/r03_ref_vs_port.txt: the port is the faster "
                  f"of the two, i.e. this baseline flatters the CPU)",
        "ms_per_filter_step": 1e3 * dt / steps,
        "one_thread": {"value": one_thread, "steps": k1, "ms_per_filter_step": 1e3 * dt1 / k1},
        "thread_sweep_particle_steps_per_s": sweep,
        "range_over_sweep": [min(v["min"] for v in sweep.values()), max(v["max"] for v in sweep.values())],
        "host": host,
    }


def pmc_traffic(kernel_substr, workload, dtype_name, t_len=20, n_override=None):
    """HBM bytes per launch of one kernel from rocprofv3 PMC counters, collected as MI355X_MICROARCH.md (HBM section)
    prescribes: FETCH_SIZE and WRITE_SIZE in *separate* --pmc passes (TCC slots), kernel-trace only; both are in KiB;
    on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read, so it is doubled
    (WRITE_SIZE is uncalibrated and taken as is).  Returns None when the profiler is unavailable."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    if shutil.which("rocprofv3") is None:
        return None
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="pf_pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "p", "--",
               sys.executable, os.path.abspath(__file__), "--_inner", "--workload", workload, "--dtype", dtype_name,
               "--T", str(t_len), "--steps", "1", "--warmup", "1", "--no-cpu-baseline"] + (["--N", str(n_override)] if n_override else [])
        env = dict(os.environ, TMPDIR="/tmp")  # (the child - `--_inner` - issues its launches directly: one dispatch row each)
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, timeout=240, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            tot, n = 0.0, 0
            for f in files:
                for row in csv.DictReader(open(f)):
                    if kernel_substr in row["Kernel_Name"] and row["Counter_Name"] == counter:
                        tot += float(row["Counter_Value"])
                        n += 1
            if n == 0:
                return None
            vals[counter] = tot / n
        except Exception:
            return None
        finally:
            shutil.rmtree(out, ignore_errors=True)
    return {"bytes_per_launch": (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0,
            "FETCH_SIZE_KiB_raw": vals["FETCH_SIZE"], "WRITE_SIZE_KiB_raw": vals["WRITE_SIZE"],
            "note": "FETCH_SIZE doubled (gfx950 wide-read correction); counters see fabric requests, so reads served "
                    "by the per-XCD L2 (the whole working set of this workload fits in L2 + Infinity Cache) do not appear"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: apf_lgo_1m (BASELINE configs[1]) - with N > 1 one replica per GPU (weak scaling: the N = 1 "
                         "job on every GPU) and the SMC^2 job (configs[4]) attached to the same line, strong and weak")
    ap.add_argument("--scaling", default=None, choices=["strong", "weak"],
                    help="--workload smc2: 1 024 theta-particles split over the GPUs (strong, default) or 1 024 per GPU (weak)")
    ap.add_argument("--no-smc2", action="store_true", help="N > 1 default workload: skip the attached SMC^2 records")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--T", type=int, default=None, help="override the number of observations")
    ap.add_argument("--N", type=int, default=None, help="override the number of particles (development: shape studies)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 PMC passes that fill roofline.traffic")
    ap.add_argument("--_inner", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--direct", action="store_true", help=argparse.SUPPRESS)  # (A/B: every run through the direct driver)
    args = ap.parse_args()
    if args.direct:
        from pyfilter_amd.hints import HINTS

        HINTS.direct = True
    if args._inner:  # profiled child of pmc_traffic(): no hipGraph replays (per-dispatch counter rows)
        from pyfilter_amd.hints import HINTS

        HINTS.graph = False

    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become N ranks, one per GPU (what the driver does for N > 1)
        import socket
        import subprocess

        share = os.environ.get("PF_BENCH_SHARE_GPU", "0") == "1"
        have = torch.cuda.device_count()
        if have < args.gpus and not share:
            sys.exit(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible (one process per GPU over RCCL)")
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")))

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # development (tests on a one-GPU box): every rank on GPU 0, collectives over gloo - RCCL refuses two ranks per device
    share = os.environ.get("PF_BENCH_SHARE_GPU", "0") == "1"
    torch.cuda.set_device(0 if share else local_rank)
    device = torch.device("cuda", 0 if share else local_rank)
    if world > 1:
        import torch.distributed as dist

        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)  # "nccl" is RCCL on ROCm
    dtype = {"f32": torch.float32, "f64": torch.float64}[args.dtype]
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); reporting n_gpus = {world}", file=sys.stderr)
    default_job = args.workload is None
    if default_job:
        # ONE workload at every N, so that value(N) / (N value(1)) is a scaling curve: BASELINE configs[1], whose single
        # filter does not shard (replicas only) - one replica per GPU, log-likelihoods all-gathered.  The
        # job that DOES shard - SMC^2, configs[4] - rides on the same line at N > 1 (``smc2_scaling``: strong and weak, each
        # with its own one-GPU point measured in the same process).
        args.workload = "apf_lgo_1m"

    import __graft_entry__ as ge

    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()

    if args.workload == "smc2":
        out = smc2_line(args, dtype, device, world, rank, args.scaling or "strong")
        if rank == 0:
            print(json.dumps(out))
        if world > 1:
            dist.destroy_process_group()
        return

    filt, y, w = build_problem(args.workload, dtype, device, world, rank, args.T, args.N)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def one_pass():
        res = filt.batch_filter(y, bar=False)
        ll = res.loglikelihood.reshape(-1)
        if world > 1:  # the path's only exchange: every rank learns every filter's log-likelihood
            if share:
                parts = [torch.empty_like(ll) for _ in range(world)]
                dist.all_gather(parts, ll.contiguous())
                return res, torch.cat(parts)
            out = torch.empty(world * ll.numel(), dtype=ll.dtype, device=device)
            dist.all_gather_into_tensor(out, ll.contiguous())
            return res, out
        return res, ll

    # set-up, not measurement: a configuration's first call allocates its plan and launches directly, its second captures
    # the hipGraph every later call replays - done before the W warm-up steps whatever W is
    for _ in range(2):
        one_pass()
    for _ in range(args.warmup):
        one_pass()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res, ll_all = one_pass()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = tmax.item()

    units_per_pass = w["N"] * w["B"] * w["T"] * world
    value = units_per_pass * args.steps / elapsed
    ll_check = check_loglikelihood(args.workload, w, ll_all)  # (raises: no throughput line for a wrong answer)

    if args.direct:
        from pyfilter_amd.hints import HINTS

        HINTS.direct = True
    if args._inner:  # profiled child of pmc_traffic(): the timed passes above are all it needs
        return

    # ---- the step kernel's duration: HIP events on the launch stream around the T launches of an instrumented pass ----
    filt._time_kernels = True
    filt.batch_filter(y, bar=False)
    torch.cuda.synchronize()
    filt._time_kernels = False
    step_ms = filt.kernel_ms[2]  # one kernel per time step: launch-to-launch duration inside the sequence
    esz = 8 if dtype == torch.float64 else 4
    bm = byte_models(w, esz)
    units = w["N"] * w["B"]  # particles one launch processes
    gbs = {k: v * units / (step_ms * 1e-3) / 1e9 for k, v in bm.items()}
    roofline = {
        # what limits the kernel at this shape by the counters (rocprofv3 --pmc SQ_INSTS_VALU per stage, profiles/r04_step_kernel_
        # pmc_stages.txt: 1 519 VALU instructions per wave at four waves per SIMD - issue-bound in every stage) ...
        "bound": "valu" if gbs["as_built"] / HBM_PEAK_GBS < 0.30 else "hbm",
        # ... while `achieved` / `peak` / `frac` stay what the bench contract defines: algorithmic bytes against the HBM peak
        # (no dense contraction on this path: MFMA does not apply)
        "priced_against": "hbm",
        "limited_by": "VALU issue + dependent-load latency inside one resident wave of workgroups, and the launch boundary - "
                      "not HBM bandwidth (the as_built bytes would take 3.1 us at peak)",
        "kernel": "k_fused_step", "achieved": gbs["survey_8d"],
        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs["survey_8d"] / HBM_PEAK_GBS, "traffic": None,
        "byte_model": "survey_8d: SURVEY.md 8(d) algorithmic bytes per particle-step (the reference dataflow's compulsory "
                      "traffic) x particles per launch / the step kernel's in-sequence duration",
        "bytes_per_particle": bm,
        "bytes_per_launch": {k: v * units for k, v in bm.items()},
        "kernel_us": {"k_fused_step": 1e3 * step_ms},
        "duration_source": "HIP events on the launch stream around the T step launches (pf_filter_run_timed) / T; the "
                           "rocprofv3 --kernel-trace --stats average of the same command is under profiles/",
        "as_built": {"achieved": gbs["as_built"], "frac": gbs["as_built"] / HBM_PEAK_GBS,
                     "note": "bytes the single kernel of a step must move as built (16 + 8 D in fp32); compare `traffic` with "
                             "bytes_per_launch.as_built"},
        "whole_job_GBs": {k: v * value / world / 1e9 for k, v in bm.items()},
    }
    if rank == 0:  # the vendor peak and a ceiling measured on this very GPU, both
        ceiling = device_copy_ceiling(device)
        roofline["measured_copy_ceiling"] = {"GB/s": ceiling, "what": "out-of-place device copy of 1 GiB (read + written bytes / HIP-event time)",
                                             "frac_of_it": {k: v / ceiling for k, v in gbs.items()}}
    kname = {"step": "k_fused_step"}
    dom = "step"

    if rank == 0 and world == 1 and not args.no_traffic:
        tr = pmc_traffic(kname[dom], args.workload, args.dtype, n_override=args.N)
        if tr is not None:
            roofline["traffic"] = tr["bytes_per_launch"]
            roofline["traffic_detail"] = tr

    if rank == 0:
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(args.workload, w)
        out = {
            "metric": "particle-steps/sec (batch x particles x T), 1M-particle APF",
            "value": value,
            "unit": "particle-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "world_size": world,
            "rccl_version": _rccl_version(world),
            "config": {
                "workload": f"{args.workload}: {w['filter'].upper()} + {w['proposal']} proposal, {w['resampler']} "
                            f"resampling, N={w['N']} particles x B={w['B']} filters per GPU, T={w['T']}, state dim {w['D']}",
                "ms_per_filter_step": 1e3 * elapsed / args.steps / w["T"],
                "parallelism": "independent filters per GPU + all-gather of log-likelihoods" if world > 1 else "single GPU",
                "loglikelihood_sample": float(ll_all.reshape(-1)[0]),
                "loglikelihood_check": ll_check,
            },
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        if cpu:
            out["speedup_vs_cpu_baseline"] = value / cpu["value"]
    if world > 1 and default_job and not args.no_smc2:
        # the sharded job of BASELINE configs[4] on the same ranks (every rank takes part; rank 0 keeps the records)
        attached = {}
        for scaling in ("strong", "weak"):
            try:
                rec = smc2_line(args, dtype, device, world, rank, scaling, attach=True)
            except Exception as exc:  # (a failure here must not cost the line its measured headline)
                rec = {"error": f"{type(exc).__name__}: {exc}"}
            attached[scaling] = rec
        if rank == 0:
            out["smc2_scaling"] = attached
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
