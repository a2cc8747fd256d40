#!/usr/bin/env python
"""The 8-GPU scaling run, rehearsed on ONE GPU: everything the driver's ``bench.py --gpus N`` does at N > 1 - launcher, rank set-up,
barriers, the sharded SMC^2 job at its full length, the JSON line - with two ranks sharing the box's GPU over gloo
(``PF_BENCH_SHARE_GPU=1``: RCCL refuses two ranks on one device; nothing else changes), so that the first real SCALE record is not
also the first full-length run.

    gpurun -- 'python tools/scale_preflight.py'            # full length (T = 500): ~3 min
    gpurun -- 'python tools/scale_preflight.py --quick'    # T = 40: the test suite's form

Checks
  1. ``bench.py --gpus 2 --workload smc2`` (BASELINE configs[4], strong scaling: 512 theta-particles per rank): one JSON line from rank
     0 with the contract's keys, ``rccl_version`` null under gloo, ``roofline`` / ``single_gpu_same_workload`` present and sane;
  2. ``bench.py --gpus 2`` (the driver's default command at N > 1): the weak-scaling line of BASELINE configs[1] with the SMC^2
     records attached (``smc2_scaling.strong / .weak``);
  3. the per-rank shapes of N = 1, 2, 4, 8 (1 024 / 512 / 256 / 128 theta-particles x 8 192): the launch trace shows the
     column-cluster kernel (SPEC 10) exactly where ``HINTS.cluster_takes`` says so, and the log-likelihoods are finite;
  4. sharded == unsharded: the SMC^2 example over two ranks gives every rank the same normalised theta-weights, equal to 1e-5 to a
     sharded run's own re-gathered weights and summing to one (``tests/test_distributed_gpu.py::_worker``).
Exit code 0 = all passed."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "world_size", "rccl_version")


def _bench(extra, timeout):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PF_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    t0 = time.perf_counter()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, cwd=ROOT, env=env, capture_output=True, text=True,
                         timeout=timeout)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert out.returncode == 0, out.stderr[-3000:]
    assert len(lines) == 1, f"rank 0 must print ONE line, got {len(lines)}"
    return json.loads(lines[0]), time.perf_counter() - t0


def check_smc2_line(quick):
    t_args = ["--T", "40"] if quick else []
    rec, wall = _bench(["--gpus", "2", "--workload", "smc2", "--steps", "1", "--warmup", "1", "--no-traffic", "--no-cpu-baseline"] + t_args,
                       1800)
    missing = [k for k in CONTRACT if k not in rec]
    assert not missing, f"keys missing from the line: {missing}"
    assert rec["n_gpus"] == 2 and rec["world_size"] == 2 and rec["steps"] == 1 and rec["warmup"] == 1
    assert rec["scaling"] == "strong" and rec["config"]["theta_per_rank"] == 512 and rec["config"]["T"] == (40 if quick else 500)
    assert rec["rccl_version"] is None, "two ranks on one GPU talk over gloo: no RCCL version to report"
    assert rec["value"] > 0 and rec["unit"] == "particle-steps/s" and rec["higher_is_better"] is True
    assert 1e-3 * rec["ms_per_step"] <= wall, "the timed region does not fit the run's wall clock"
    roof, solo = rec["roofline"], rec["single_gpu_same_workload"]
    assert roof and "error" not in roof and 0 < roof["frac"] <= 1.2, roof
    assert solo and "error" not in solo and solo["value"] > 0, solo
    return f"smc2 line: {rec['value']:.3e} particle-steps/s, {rec['ms_per_step']:.1f} ms per fit; one GPU alone {solo['value']:.3e}; roofline {roof['frac']:.2f}"


def check_default_line(quick):
    t_args = ["--T", "24", "--N", "65536"] if quick else []
    rec, _ = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-traffic"] + t_args, 2400)
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["config"]["workload"].startswith("apf_lgo_1m")
    assert rec["rccl_version"] is None and rec["roofline"]["frac"] > 0
    strong, weak = rec["smc2_scaling"]["strong"], rec["smc2_scaling"]["weak"]
    for r, per in ((strong, 512), (weak, 1024)):
        assert "error" not in r and r["config"]["theta_per_rank"] == per and r["value"] > 0, r
        assert r["single_gpu_same_workload"]["value"] > 0 and r["roofline"]["frac"] > 0
    return (f"default line: {rec['value']:.3e} particle-steps/s (weak, 2 replicas); smc2 strong {strong['value']:.3e}, "
            f"weak {weak['value']:.3e}")


def check_per_rank_routes():
    import torch

    from pyfilter_amd import ops, resampling, timeseries as ts
    from pyfilter_amd.filters.particle import APF, proposals
    from pyfilter_amd.hints import HINTS
    from pyfilter_amd.timeseries import models

    dev = "cuda"
    t = lambda v: torch.tensor(v, dtype=torch.float32, device=dev)  # noqa: E731
    y = (0.05 * torch.randn(12)).cumsum(0).to(dev)
    notes = []
    for ranks in (1, 2, 4, 8):
        b = 1024 // ranks
        kappa = torch.linspace(0.02, 0.2, b, device=dev)
        ssm = ts.LinearStateSpaceModel(models.OrnsteinUhlenbeck(kappa, torch.zeros(b, device=dev), 0.05 + torch.zeros(b, device=dev)), (t(1.0), t(0.05)))
        f = APF(ssm, 8192, proposal=proposals.LinearGaussianObservations(), resampling=resampling.systematic, seed=ranks)
        f.set_batch_shape(torch.Size([b]))
        res = f.batch_filter(y, bar=False)
        torch.cuda.synchronize()
        spec = ops.debug_launch_trace(1)[-1]["SPEC"]
        takes = HINTS.cluster_takes(8192, b)
        assert (spec == 10) == takes, f"{ranks} rank(s): {b} x 8192 ran SPEC {spec}, cluster_takes says {takes}"
        assert torch.isfinite(res.loglikelihood).all() and getattr(f, "cluster_fallbacks", 0) == 0
        notes.append(f"N={ranks}: {b} x 8192 -> {'cluster' if takes else 'per-step'}")
    return "per-rank routes: " + "; ".join(notes)


def check_sharded_equals_unsharded(tmp):
    import torch
    import torch.multiprocessing as mp

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from tests.test_distributed_gpu import _free_port, _worker

    out = os.path.join(tmp, "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out, 8), nprocs=2, join=True)
    got = torch.load(out)
    assert got["same"], "the ranks disagree on the theta-weights"
    assert got["local_theta"] == 48 and abs(got["w"].sum().item() - 1.0) < 1e-5 and torch.isfinite(got["ll"]).all()
    assert got["moves"] >= 1, "no rejuvenation: the redistribution of whole filters was not exercised"
    return f"two ranks: identical theta-weights on both, {got['moves']} PMMH moves, weights sum to 1 within 1e-5"


def main():
    import tempfile

    quick = "--quick" in sys.argv
    failed = 0
    with tempfile.TemporaryDirectory() as tmp:
        for name, fn in (("smc2 line", lambda: check_smc2_line(quick)), ("default line", lambda: check_default_line(quick)),
                         ("routes", check_per_rank_routes), ("sharded == unsharded", lambda: check_sharded_equals_unsharded(tmp))):
            t0 = time.perf_counter()
            try:
                print(f"PASS  {fn()}  [{time.perf_counter() - t0:.0f} s]", flush=True)
            except Exception as exc:  # noqa: BLE001
                failed += 1
                print(f"FAIL  {name}: {type(exc).__name__}: {exc}", flush=True)
    print("scale preflight:", "ok" if not failed else f"{failed} check(s) failed")
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
