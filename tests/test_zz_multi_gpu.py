"""Paths that need MORE THAN ONE GPU: RCCL between real devices, the device update log and the
asynchronous shards across xGMI. They skip on a one-GPU box (where the same protocols are covered by
ranks sharing the device, tests/test_update_log_gpu.py); the file sorts last on purpose."""
import json
import os
import subprocess
import sys
import time

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
    from distributed_tensorflow_amd.parallel.strategies import Backup, FullSync, Interval, SoftKill
    from distributed_tensorflow_amd.train.replica import HipReplica

    comm = Communicator.from_env()                       # nccl = RCCL, one device per rank
    board = control_plane.open_board(board_name, world, rank)
    rep = HipReplica("resnet", 32, num_residual_blocks=1, abortable=mode in ("backup", "soft_kill"))
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
        if mode == "backup":
            strat = Backup(comm, board, world - 1)
        elif mode == "interval":
            strat = Interval(comm, board, 3.0)
        elif mode == "soft_kill":
            strat = SoftKill(comm, board, world - 1)
        else:
            if mode == "p2p":
                comm.enable_p2p(rep.arena.grads_full.numel())   # the hand-written two-shot all-reduce over xGMI
            strat = FullSync(comm, board, reduce_scatter=(mode == "sharded"))
        for it in range(24 if mode == "interval" else 8):
            strat.step(rep, it, 1e-3)
            losses.append(rep.loss_value())
            if mode == "interval":
                time.sleep(0.004)
        strat.finish()
        rep.synchronize()
        extra = dict(updates=rep.updates_applied)
    out = dict(losses=losses, psum=rep.arena.params.double().sum().item(), device=torch.cuda.current_device(),
               topo=comm.topology() if hasattr(comm, "topology") else {}, **extra)
    comm.barrier()
    comm.close()
    return out


@pytest.mark.parametrize("mode", ["sync", "sharded", "p2p", "backup", "interval", "soft_kill", "async"])
def test_two_gpus(mode):
    out = run_workers(_worker, 2, f"dtf_mg_{mode}_{os.getpid()}", mode, timeout=300)
    assert {o["device"] for o in out} == {0, 1}
    assert abs(out[0]["psum"] - out[1]["psum"]) <= 1e-9 * max(1.0, abs(out[0]["psum"]))
    if mode in ("sync", "sharded", "p2p"):
        assert all(o["updates"] == 8 for o in out) and out[0]["losses"][-1] < out[0]["losses"][0]
    elif mode == "backup":
        assert out[0]["updates"] == out[1]["updates"] and 8 <= out[0]["updates"] <= 16
    elif mode == "interval":
        assert out[0]["updates"] == out[1]["updates"] >= 2          # the timer's windows, the same on both
    elif mode == "soft_kill":
        assert all(o["updates"] == 8 for o in out)
    else:
        assert out[0]["versions"] == [16, 16]


def _probe_worker(rank, world, board_name):
    from distributed_tensorflow_amd.parallel.comm import Communicator
    from distributed_tensorflow_amd.parallel.p2p import link_probe

    comm = Communicator.from_env()
    out = link_probe(comm, payload_bytes=1 << 20, iters=200, cas_iters=5000)
    out["device"] = torch.cuda.current_device()
    comm.close()
    return out


def test_link_probe_between_two_gpus():
    """First contact over xGMI with the update log's own primitives: a flag ping-pong (latency), a flag
    + 1 MB ping-pong (payload intact behind a release / acquire pair; one-workgroup bandwidth) and a
    compare-exchange counter contended from both GPUs. The numbers are printed for the bring-up log."""
    out = run_workers(_probe_worker, 2, f"dtf_mg_probe_{os.getpid()}", timeout=300)
    print("link probe:", json.dumps(out))
    assert {o["device"] for o in out} == {0, 1}
    for o in out:
        assert not o["timed_out"] and o["bad_words"] == 0 and o["cas_lost"] == 0, o
        assert o["flag_round_trip_us"] < 200 and o["payload_gb_s"] > 1.0, o


def test_bench_launches_one_rccl_rank_per_gpu():
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "10", "--warmup", "3", "--batch", "128"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["ranks"] == 2 and d["distinct_gpus"] == 2 and d["rccl_ranks"] == 2 and d["backend"] == "nccl"
    assert d["config"]["global_batch"] == 256
