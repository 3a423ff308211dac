"""Paths that need MORE THAN ONE GPU: RCCL between real devices, the device update log and the
asynchronous shards across xGMI. They skip on a one-GPU box (where the same protocols are covered by
ranks sharing the device, tests/test_update_log_gpu.py); the file sorts last on purpose."""
import json
import os
import subprocess
import sys

import pytest
import torch

from mp_util import run_workers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2, reason="needs at least two GPUs")]


def _worker(rank, world, board_name, mode):
    from distributed_tensorflow_amd.parallel import control_plane
    from distributed_tensorflow_amd.parallel.async_ps import AsyncParameterShards
    from distributed_tensorflow_amd.parallel.comm import Communicator
    from distributed_tensorflow_amd.parallel.strategies import Backup, FullSync
    from distributed_tensorflow_amd.train.replica import HipReplica

    comm = Communicator.from_env()                       # nccl = RCCL, one device per rank
    board = control_plane.open_board(board_name, world, rank)
    rep = HipReplica("resnet", 32, num_residual_blocks=1, abortable=mode == "backup")
    g = torch.Generator().manual_seed(50 + rank)
    rep.load_batch(torch.randn(32, 32, 32, 3, generator=g), torch.randint(0, 10, (32,), generator=g))
    control_plane.launch_manager(board, rank, world, 120)
    with rep.stream_ctx():
        comm.broadcast(rep.arena.params, 0)
    rep.synchronize()
    losses = []
    if mode == "async":
        shards = AsyncParameterShards(rep, comm, board, 2 * world, board_name + "_ps", 1e-3)
        losses = [shards.step()[0] for _ in range(8)]
        shards.drain(60)
        rep.synchronize()
        extra = dict(versions=shards.versions(), applied=shards.stats.applied)
        shards.close()
    else:
        strat = Backup(comm, board, world - 1) if mode == "backup" else FullSync(comm, board, reduce_scatter=(mode == "sharded"))
        for it in range(8):
            strat.step(rep, it, 1e-3)
            losses.append(rep.loss_value())
        strat.finish()
        rep.synchronize()
        extra = dict(updates=rep.updates_applied)
    out = dict(losses=losses, psum=rep.arena.params.double().sum().item(), device=torch.cuda.current_device(),
               topo=comm.topology() if hasattr(comm, "topology") else {}, **extra)
    comm.barrier()
    comm.close()
    return out


@pytest.mark.parametrize("mode", ["sync", "sharded", "backup", "async"])
def test_two_gpus(mode):
    out = run_workers(_worker, 2, f"dtf_mg_{mode}_{os.getpid()}", mode, timeout=300)
    assert {o["device"] for o in out} == {0, 1}
    assert abs(out[0]["psum"] - out[1]["psum"]) <= 1e-9 * max(1.0, abs(out[0]["psum"]))
    if mode in ("sync", "sharded"):
        assert all(o["updates"] == 8 for o in out) and out[0]["losses"][-1] < out[0]["losses"][0]
    elif mode == "backup":
        assert out[0]["updates"] == out[1]["updates"] and 8 <= out[0]["updates"] <= 16
    else:
        assert out[0]["versions"] == [16, 16]


def test_bench_launches_one_rccl_rank_per_gpu():
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "10", "--warmup", "3", "--batch", "128"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["ranks"] == 2 and d["distinct_gpus"] == 2 and d["rccl_ranks"] == 2 and d["backend"] == "nccl"
    assert d["config"]["global_batch"] == 256
