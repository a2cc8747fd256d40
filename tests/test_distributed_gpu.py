"""The sharded SMC^2 driver on the real HIP filters: two processes share the one GPU of the test box, collectives over
``gloo`` (RCCL refuses two ranks on one device; on a multi-GPU node the same code runs one rank per GPU over RCCL).  Every
rank owns half of the theta-particles; the theta-weights are all-gathered per observation and a rejuvenation redistributes
whole filters (``Shard.take`` on the device buffers behind ``FilterResult``)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _data(t_len, seed=1, beta=0.8, sigma=0.4):
    g = torch.Generator().manual_seed(seed)
    x, ys = 0.0, []
    for _ in range(t_len):
        x = beta * x + sigma * torch.randn((), generator=g).item()
        ys.append(x + 0.3 * torch.randn((), generator=g).item())
    return torch.tensor(ys)


def _worker(rank, world, port, out, block):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import importlib.util

        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        spec = importlib.util.spec_from_file_location("smc2_example", os.path.join(root, "examples", "smc2_linear_gaussian.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        res = mod.smc2(_data(80).cuda(), n_theta=96, n_state=1024, ess_frac=0.5, seed=3, block=block)
        w = res["weights"]            # normalised weights of ALL theta-particles: must be identical on every rank
        gathered = [torch.empty_like(w) for _ in range(world)]
        dist.all_gather(gathered, w)
        same = all(torch.equal(gathered[0], g_) for g_ in gathered)
        if rank == 0:
            torch.save({"mean": res["mean"].cpu(), "moves": res["moves"], "same": same, "w": w.cpu(),
                        "local_theta": res["theta"].batch_shape[0], "ll": res["loglikelihood"].cpu()}, out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("block", [1, 8])
def test_smc2_two_ranks_on_the_hip_filters(tmp_path, block):
    """``block = 1``: observation by observation (one all-gather of B weights each); ``block = 8``: ``fit`` with the filters
    running ahead of the rejuvenation test (one all-gather of the block's ``(8, B)`` weight paths)."""
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out, block), nprocs=2, join=True)
    got = torch.load(out)
    assert got["same"], "the ranks disagree on the theta-weights"
    assert got["local_theta"] == 48 and got["w"].shape == (96,) and abs(got["w"].sum().item() - 1.0) < 1e-5
    assert got["moves"] >= 1, "no rejuvenation happened: the redistribution path was not exercised"
    b, s = got["mean"].tolist()
    assert abs(b - 0.8) < 0.2 and abs(s - 0.4) < 0.15, (b, s)
    assert torch.isfinite(got["ll"]).all()


@pytest.mark.parametrize("workload,extra", [("apf_lgo_1m", ["--T", "12", "--N", "65536"]), ("smc2", ["--T", "30"])])
def test_bench_under_torchrun_with_two_ranks(workload, extra):
    """``bench.py`` the way the driver launches it for N > 1 (``python -m torch.distributed.run --nproc-per-node N``): the
    rank-0 build + barrier, the barrier-bracketed timed region, the max-over-ranks reduction and the one JSON line.  Both
    ranks share the test box's single GPU (``PF_BENCH_SHARE_GPU=1``: gloo instead of RCCL, nothing else changes)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PF_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--workload", workload, "--no-traffic", "--no-cpu-baseline"] + extra
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]  # rank 0 prints, once
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["world_size"] == 2 and rec["steps"] == 2 and rec["warmup"] == 1
    assert rec["value"] > 0 and rec["unit"] == "particle-steps/s" and rec["higher_is_better"] is True
    assert rec["scaling"] == ("strong" if workload == "smc2" else "weak")


def test_bench_spawns_its_own_ranks_one_workload_at_every_n():
    """``python bench.py --gpus 2`` with no launcher around it (the shape of the driver's N = 1 command): the script
    re-executes itself under ``torch.distributed.run`` with two ranks.  The default workload is the N = 1 one (BASELINE
    configs[1], one replica per GPU: weak scaling - ``value(N) / (N value(1))`` is a scaling curve), and the job that shards -
    SMC^2, configs[4] - rides on the same line, strong and weak, each with its own one-GPU point and a roofline object."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PF_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--T", "24", "--N", "65536"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["world_size"] == 2 and rec["scaling"] == "weak"
    assert rec["config"]["workload"].startswith("apf_lgo_1m") and rec["value"] > 0 and rec["roofline"]["frac"] > 0
    strong, weak = rec["smc2_scaling"]["strong"], rec["smc2_scaling"]["weak"]
    assert "error" not in strong and "error" not in weak, (strong, weak)
    assert strong["scaling"] == "strong" and strong["config"]["theta_per_rank"] == 512 and strong["value"] > 0
    assert weak["scaling"] == "weak" and weak["config"]["theta_per_rank"] == 1024 and weak["value"] > 0
    for r in (strong, weak):
        assert r["single_gpu_same_workload"]["value"] > 0 and r["roofline"]["kernel"] == "k_fused_step" and r["roofline"]["frac"] > 0


def test_bench_smc2_is_one_workload_from_one_gpu_on():
    """``bench.py --gpus 1 --workload smc2`` is the N = 1 point of the SMC^2 curves (strong == weak there) and carries the
    ``roofline`` of its dominant kernel at the per-rank shape and a ``cpu_baseline`` like every N = 1 line."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--workload", "smc2", "--steps", "1", "--warmup", "1", "--T", "40"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert rec["n_gpus"] == 1 and rec["scaling"] == "strong" and rec["config"]["theta_per_rank"] == 1024
    assert rec["roofline"]["kernel"] == "k_fused_step" and 0 < rec["roofline"]["frac"] < 1.5
    assert rec["cpu_baseline"]["kind"] == "port" and rec["cpu_baseline"]["value"] > 0 and rec["speedup_vs_cpu_baseline"] > 1


def _rccl_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    device = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)  # "nccl" is RCCL on ROCm
    try:
        from pyfilter_amd.distributed import Shard

        total = 64 * world
        sh = Shard(total)
        g = torch.Generator().manual_seed(11)
        full = torch.randn(total, 4096, generator=g)
        planes = torch.randn(3, total, 512, generator=g)
        idx = torch.randint(0, total, (total,), generator=g)
        mine = sh.slice(idx).to(device)
        # even blocks on RCCL: all_gather_into_tensor straight into the concatenated tensor
        got = sh.all_gather(sh.slice(full).to(device))
        ok = torch.equal(got.cpu(), full)
        got1 = sh.all_gather(sh.slice(planes, dim=1).to(device), dim=1)
        ok &= torch.equal(got1.cpu(), planes)
        route = sh.route(mine)  # all_to_all_single over xGMI: only the columns that change owner
        ok &= torch.equal(route.take(sh.slice(full).to(device)).cpu(), full[sh.slice(idx)])
        ok &= torch.equal(route.take(sh.slice(planes, dim=1).to(device), dim=1).cpu(), planes[:, sh.slice(idx)])
        ok &= sh.all_max(torch.tensor([float(rank)], device=device)).item() == world - 1
        ok &= abs(sh.all_mean(torch.tensor(float(rank + 1), device=device), 2).item() - (world + 1) / 4) < 1e-6
        flags = [None] * world
        dist.all_gather_object(flags, bool(ok))
        if rank == 0:
            torch.save({"ok": all(flags), "rccl": torch.cuda.nccl.version(), "moved": route.moved}, out)
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank: runs on multi-GPU nodes only")
def test_rccl_collectives_of_the_sharded_driver(tmp_path):
    """The three exchanges of the sharded SMC^2 driver over RCCL / xGMI, one rank per GPU: the ``all_gather_into_tensor``
    fast path of ``Shard.all_gather``, ``Route.take`` (``all_to_all_single`` with uneven splits) and the small
    all-reduces - against plain indexing of the same global data."""
    world = min(torch.cuda.device_count(), 8)
    out = str(tmp_path / "r0.pt")
    mp.spawn(_rccl_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = torch.load(out)
    assert got["ok"], got


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank: runs on multi-GPU nodes only")
def test_bench_smc2_over_rccl():
    """``python bench.py --gpus N`` on a multi-GPU node: N ranks over RCCL, theta-particles sharded."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = min(torch.cuda.device_count(), 8)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PF_BENCH_SHARE_GPU")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "1", "--T", "60"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert rec["n_gpus"] == n and rec["rccl_version"]
    assert rec["smc2_scaling"]["strong"]["config"]["theta_per_rank"] == 1024 // n and rec["smc2_scaling"]["weak"]["config"]["theta_per_rank"] == 1024


def _rccl_world_of_one(_rank, port, out):
    """RCCL with ONE rank on the box's one GPU, every exchange of the sharded driver FORCED through its collective branch
    (``distributed.force_collectives``): the exact ``torch.distributed`` calls of an N-GPU job - ``all_gather_into_tensor``
    (even blocks on nccl), ``all_to_all_single`` with split lists, the small all-reduces - issued on RCCL, and a whole SMC^2
    fit through them, against the same fit with no process group in the way (``SOLO``)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)  # "nccl" is RCCL on ROCm
    try:
        from pyfilter_amd import distributed as D

        D.force_collectives(True)
        sh = D.Shard(96)
        assert sh.collective and sh.world == 1 and dist.get_backend() == "nccl"
        g = torch.Generator().manual_seed(5)
        full = torch.randn(96, 2048, generator=g).to(device)
        planes = torch.randn(3, 96, 512, generator=g).to(device)
        idx = torch.randint(0, 96, (96,), generator=g).to(device)
        ok = torch.equal(sh.all_gather(full), full)                       # all_gather_into_tensor
        ok &= torch.equal(sh.all_gather(planes, dim=1), planes)
        route = sh.route(idx)                                             # all-gather of the wants + host plan
        ok &= torch.equal(route.take(full), full[idx])                    # all_to_all_single (empty splits: nothing changes owner)
        ok &= torch.equal(route.take(planes, dim=1), planes[:, idx])
        ok &= sh.all_max(torch.tensor([3.0], device=device)).item() == 3.0    # all_reduce MAX
        ok &= abs(sh.all_mean(torch.tensor(6.0, device=device), 4).item() - 1.5) < 1e-6   # all_reduce SUM
        # the algorithm through the same calls: theta-weights all-gathered per block, a rejuvenation's routed redistribution
        import importlib.util

        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        spec = importlib.util.spec_from_file_location("smc2_example", os.path.join(root, "examples", "smc2_linear_gaussian.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        y = _data(60).cuda()
        forced = mod.smc2(y, n_theta=64, n_state=512, ess_frac=0.5, seed=3, block=8)
        D.force_collectives(False)
        plain = mod.smc2(y, n_theta=64, n_state=512, ess_frac=0.5, seed=3, block=8)
        torch.save({"ok": bool(ok), "rccl": torch.cuda.nccl.version(), "moves": (forced["moves"], plain["moves"]),
                    "w": (forced["weights"].cpu(), plain["weights"].cpu()), "mean": (forced["mean"].cpu(), plain["mean"].cpu()),
                    "ll": (forced["loglikelihood"].cpu(), plain["loglikelihood"].cpu())}, out)
    finally:
        dist.destroy_process_group()


def test_rccl_world_of_one_runs_every_exchange_of_the_sharded_driver(tmp_path):
    """Runs on the one-GPU test box: the RCCL library is loaded, a communicator is built and the driver's collectives execute
    on it (a world of one - what a one-GPU box can offer); the forced fit must land on the plain one's numbers (a one-rank
    all-gather / all-to-all returns its input)."""
    out = str(tmp_path / "r0.pt")
    mp.spawn(_rccl_world_of_one, args=(_free_port(), out), nprocs=1, join=True)
    got = torch.load(out)
    assert got["ok"], got
    assert got["moves"][0] == got["moves"][1] and got["moves"][0] >= 1, got["moves"]
    for k in ("w", "mean", "ll"):  # (the sharded branches evaluate the theta-level arithmetic with torch ops where the one-rank
        # route takes one kernel: same numbers to rounding)
        torch.testing.assert_close(got[k][0], got[k][1], rtol=1e-5, atol=1e-6, msg=k)


def test_scale_preflight_quick():
    """``tools/scale_preflight.py --quick``: the one command that rehearses the driver's N > 1 bench on a one-GPU box (JSON contract of
    both lines, per-rank kernel routes, sharded == unsharded) - at full length it is what a round runs before its SCALE record."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "scale_preflight.py"), "--quick"], cwd=root, env=env,
                         capture_output=True, text=True, timeout=1800)
    assert out.returncode == 0 and "scale preflight: ok" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
