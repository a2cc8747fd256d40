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
