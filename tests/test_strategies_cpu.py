"""Aggregation-policy semantics on CPU: several local processes joined by gloo + the shared-memory
control board, with injected stragglers (the reference's own research scenario)."""
import os
import time

import numpy as np
import torch

from mp_util import run_workers

BATCH = 8


def _setup(rank, world, board_name, seed_data=True):
    from distributed_tensorflow_amd.parallel import control_plane
    from distributed_tensorflow_amd.parallel.comm import Communicator
    from distributed_tensorflow_amd.train.replica import TorchReplica

    comm = Communicator.from_env(device="cpu")
    board = control_plane.open_board(board_name, world, rank)
    rep = TorchReplica("resnet", BATCH, num_residual_blocks=1, optimizer="adam")
    g = torch.Generator().manual_seed(100 + rank)
    images = torch.randn(BATCH, 32, 32, 3, generator=g)
    labels = torch.randint(0, 10, (BATCH,), generator=g)
    rep.load_batch(images, labels)
    control_plane.launch_manager(board, rank, world, 60)
    return comm, board, rep, images, labels


def _grads_of(rep):
    rep.forward_backward()
    return rep.arena.grads.clone()


def _full_sync_worker(rank, world, board_name, sharded=False):
    from distributed_tensorflow_amd.parallel.strategies import FullSync

    comm, board, rep, images, labels = _setup(rank, world, board_name)
    own = _grads_of(rep)
    gathered = [torch.zeros_like(own) for _ in range(world)]
    torch.distributed.all_gather(gathered, own)
    mean = torch.stack(gathered).mean(0)
    p0 = rep.arena.params.clone()
    res = FullSync(comm, board, reduce_scatter=sharded).step(rep, 0, 1e-3)
    # expected: Adam's first step on the mean gradient
    from distributed_tensorflow_amd.ops import reference as ref

    want = p0.clone()
    ref.adam_update(want, mean, torch.zeros_like(want), torch.zeros_like(want), ref.adam_lr_t(1e-3, 1))
    err = float((rep.arena.params - want).abs().max())
    comm.barrier()
    comm.close()
    return res.contributors, err, rep.arena.params.sum().item()


def test_full_sync_is_the_mean_of_all_replicas():
    out = run_workers(_full_sync_worker, 3, f"dtf_t_sync_{os.getpid()}")
    assert all(c == 3 for c, _, _ in out)
    assert all(e < 1e-6 for _, e, _ in out)
    assert len({round(s, 4) for _, _, s in out}) == 1           # identical parameters everywhere


def test_sharded_optimizer_matches_the_replicated_update():
    # reduce-scatter + per-shard Adam + all-gather must give the same parameters as all-reduce + full Adam
    out = run_workers(_full_sync_worker, 3, f"dtf_t_rs_{os.getpid()}", True)
    assert all(e < 1e-6 for _, e, _ in out)
    assert len({round(s, 4) for _, _, s in out}) == 1


def _backup_worker(rank, world, board_name, m, slow_rank):
    from distributed_tensorflow_amd.ops import reference as ref
    from distributed_tensorflow_amd.parallel.strategies import Backup

    comm, board, rep, images, labels = _setup(rank, world, board_name)
    own = _grads_of(rep)
    gathered = [torch.zeros_like(own) for _ in range(world)]
    torch.distributed.all_gather(gathered, own)
    p0 = rep.arena.params.clone()
    strat = Backup(comm, board, m)
    comm.barrier()
    if rank == slow_rank:
        time.sleep(1.0)                                      # the straggler
    t0 = time.time()
    res = strat.step(rep, 0, 1e-3)
    dt = time.time() - t0
    mask = board.mask(0)
    fast = [r for r in range(world) if r != slow_rank]
    mean = torch.stack([gathered[r] for r in fast]).mean(0)
    want = p0.clone()
    ref.adam_update(want, mean, torch.zeros_like(want), torch.zeros_like(want), ref.adam_lr_t(1e-3, 1))
    err = float((rep.arena.params - want).abs().max())
    comm.barrier()
    comm.close()
    return dict(included=res.included, contributors=res.contributors, err=err, mask=mask, dt=dt, updates=rep.updates_applied)


def test_backup_workers_drop_exactly_the_straggler_and_everyone_advances():
    world, m, slow = 3, 2, 1
    out = run_workers(_backup_worker, world, f"dtf_t_bk_{os.getpid()}", m, slow)
    assert [o["included"] for o in out] == [True, False, True]
    assert all(o["contributors"] == m for o in out)
    assert all(o["mask"] == 0b101 for o in out)               # first M arrivals, agreed by all replicas
    assert all(o["err"] < 1e-6 for o in out)                  # update = mean of the two fresh gradients
    assert all(o["updates"] == 1 for o in out)                # the straggler moved on too


def _soft_kill_worker(rank, world, board_name, k, slow_rank):
    from distributed_tensorflow_amd.parallel.strategies import SoftKill

    comm, board, rep, images, labels = _setup(rank, world, board_name)
    strat = SoftKill(comm, board, k)
    comm.barrier()
    if rank == slow_rank:
        # arrive after the others are done, so the stop flag is already up when backprop starts
        deadline = time.time() + 30
        while board.arrivals(0) < k and time.time() < deadline:
            time.sleep(0.01)
    res = strat.step(rep, 0, 1e-3)
    nz = int((rep.arena.grads != 0).sum())
    comm.barrier()
    comm.close()
    return dict(aborted=res.aborted, contributors=res.contributors, psum=rep.arena.params.sum().item())


def test_soft_kill_short_circuits_the_straggler():
    world, k, slow = 3, 2, 2
    out = run_workers(_soft_kill_worker, world, f"dtf_t_sk_{os.getpid()}", k, slow)
    assert [o["aborted"] for o in out] == [False, False, True]
    assert all(o["contributors"] == world for o in out)       # zeros are still "pushed" (mean over N)
    assert len({round(o["psum"], 4) for o in out}) == 1


def _interval_worker(rank, world, board_name, interval_ms, slow_rank):
    from distributed_tensorflow_amd.parallel.strategies import Interval

    comm, board, rep, images, labels = _setup(rank, world, board_name)
    strat = Interval(comm, board, interval_ms)
    comm.barrier()
    res0 = strat.step(rep, 0, 1e-3)                            # everyone in time
    comm.barrier()
    board.open_step(1)                                         # the window opens now for all
    if rank == slow_rank:
        time.sleep(interval_ms / 1e3 + 0.3)                    # misses the deadline
    res1 = strat.step(rep, 1, 1e-3)
    comm.barrier()
    comm.close()
    return dict(c0=res0.contributors, c1=res1.contributors, inc1=res1.included, applied=(res0.applied, res1.applied),
                psum=rep.arena.params.sum().item())


def test_interval_aggregates_what_arrived_in_time():
    world, slow = 3, 0
    out = run_workers(_interval_worker, world, f"dtf_t_iv_{os.getpid()}", 1500, slow)
    assert all(o["c0"] == world for o in out)
    assert all(o["c1"] == world - 1 for o in out)
    assert [o["inc1"] for o in out] == [False, True, True]
    assert len({round(o["psum"], 4) for o in out}) == 1


def _weighted_worker(rank, world, board_name):
    from distributed_tensorflow_amd.parallel.strategies import Weighted

    comm, board, rep, images, labels = _setup(rank, world, board_name)
    strat = Weighted(comm, board, num_worker_kill=1, random_seed=142)
    w = strat.weights_for_step(0)
    res = strat.step(rep, 0, 1e-3)
    comm.barrier()
    comm.close()
    return dict(w=w.tolist(), included=res.included, psum=rep.arena.params.sum().item())


def test_weighted_least_squares_aggregation():
    from distributed_tensorflow_amd.parallel.strategies import ls_gradient_weights

    # no kills: every batch is covered twice, so uniform weights 1/2 reproduce the plain sum
    w = ls_gradient_weights(4, 4, [])
    assert np.allclose(w, 0.5)
    # one killed worker: its weight is zero and A^T w still approximates the all-ones vector
    w = ls_gradient_weights(4, 4, [2])
    assert w[2] == 0
    a = np.zeros((4, 4))
    for r in range(4):
        a[r, r] = a[r, (r + 1) % 4] = 1
    a[2] = 0
    assert np.linalg.norm(a.T @ w - 1) <= np.linalg.norm(a.T @ np.full(4, 0.5) * (np.arange(4) != 2) - 1) + 1e-9
    out = run_workers(_weighted_worker, 3, f"dtf_t_w_{os.getpid()}")
    assert all(o["w"] == out[0]["w"] for o in out)             # same kill draw on every replica
    killed = [i for i, x in enumerate(out[0]["w"]) if x == 0]
    assert len(killed) == 1 and [o["included"] for o in out] == [i not in killed for i in range(3)]
    assert len({round(o["psum"], 4) for o in out}) == 1


def _async_worker(rank, world, board_name, tau, steps):
    from distributed_tensorflow_amd.parallel.async_ps import AsyncParameterShards

    comm, board, rep, images, labels = _setup(rank, world, board_name)
    shards = AsyncParameterShards(rep, comm, board, tau, board_name + "_ps", lambda v: 1e-3)
    losses = []
    for i in range(steps):
        if rank == 1:
            time.sleep(0.05)                                   # a slower replica: others do not wait for it
        loss, gs = shards.step()
        losses.append(loss)
    shards.drain(60)
    versions = [board.shard_version(o) for o in range(world)]
    psum = rep.arena.params.double().sum().item()
    stats = vars(shards.stats)
    shards.close()
    comm.barrier()
    comm.close()
    return dict(losses=losses, versions=versions, psum=psum, stats=stats)


def test_async_shards_apply_pushes_and_converge():
    world, steps = 3, 6
    out = run_workers(_async_worker, world, f"dtf_t_as_{os.getpid()}", 8, steps)
    # every push is either applied or dropped, none lost; with a generous bound nothing is stale
    assert sum(o["stats"]["pushed"] for o in out) == world * world * steps
    total_applied = sum(o["stats"]["applied"] for o in out)
    total_dropped = sum(o["stats"]["dropped_stale"] for o in out)
    assert total_applied + total_dropped == world * world * steps and total_dropped == 0
    # shard versions count the updates each owner applied
    assert out[0]["versions"] == [world * steps] * world
    # after the final pull every replica holds the same parameters, and training made progress
    assert len({round(o["psum"], 3) for o in out}) == 1
    assert out[0]["losses"][-1] < out[0]["losses"][0]


def test_async_staleness_bound_drops_old_gradients():
    world, steps = 2, 5
    out = run_workers(_async_worker, world, f"dtf_t_as0_{os.getpid()}", 0, steps)
    dropped = sum(o["stats"]["dropped_stale"] for o in out)
    applied = sum(o["stats"]["applied"] for o in out)
    assert applied + dropped == world * world * steps
    assert dropped > 0            # tau = 0: a gradient computed before another update landed is discarded
    assert len({round(o["psum"], 3) for o in out}) == 1
