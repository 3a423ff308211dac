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
    issued0 = comm.collectives_issued
    res = FullSync(comm, board, reduce_scatter=sharded).step(rep, 0, 1e-3)
    issued = comm.collectives_issued - issued0
    # expected: Adam's first step on the mean gradient
    from distributed_tensorflow_amd.ops import reference as ref

    want = p0.clone()
    ref.adam_update(want, mean, torch.zeros_like(want), torch.zeros_like(want), ref.adam_lr_t(1e-3, 1))
    err = float((rep.arena.params - want).abs().max())
    comm.barrier()
    comm.close()
    return res.contributors, err, rep.arena.params.sum().item(), issued


def test_full_sync_is_the_mean_of_all_replicas():
    out = run_workers(_full_sync_worker, 3, f"dtf_t_sync_{os.getpid()}")
    assert all(c == 3 for c, _, _, _ in out)
    assert all(e < 1e-6 for _, e, _, _ in out)
    assert len({round(s, 4) for _, _, s, _ in out}) == 1        # identical parameters everywhere
    assert all(n == 1 for _, _, _, n in out)                    # one all-reduce for a gradient this small


def test_sharded_optimizer_matches_the_replicated_update():
    # reduce-scatter + per-shard Adam + all-gather must give the same parameters as all-reduce + full Adam
    # (3 ranks: the arena size is not divisible by the world size, the shards reach into the padding)
    out = run_workers(_full_sync_worker, 3, f"dtf_t_rs_{os.getpid()}", True)
    assert all(e < 1e-6 for _, e, _, _ in out)
    assert len({round(s, 4) for _, _, s, _ in out}) == 1
    assert all(n == 2 for _, _, _, n in out)                    # ONE reduce-scatter + ONE all-gather per step


def _backup_worker(rank, world, board_name, m, slow_rank):
    from distributed_tensorflow_amd.ops import reference as ref
    from distributed_tensorflow_amd.parallel.strategies import Backup

    comm, board, rep, images, labels = _setup(rank, world, board_name)
    own = _grads_of(rep)
    gathered = [torch.zeros_like(own) for _ in range(world)]
    torch.distributed.all_gather(gathered, own)
    p0 = rep.arena.params.clone()
    strat = Backup(comm, board, m)
    strat._get_log(rep)                                      # set-up (collective) outside the timed step
    comm.barrier()
    if rank == slow_rank:
        time.sleep(1.5)                                      # the straggler
    t0 = time.time()
    res = strat.step(rep, 0, 1e-3)
    dt = time.time() - t0
    count, mask, owners, scale = board.agg_entry(0)
    fast = [r for r in range(world) if r != slow_rank]
    mean = torch.stack([gathered[r] for r in fast]).mean(0)
    want = p0.clone()
    ref.adam_update(want, mean, torch.zeros_like(want), torch.zeros_like(want), ref.adam_lr_t(1e-3, 1))
    err = float((rep.arena.params - want).abs().max())
    strat.finish()
    comm.close()
    return dict(included=res.included, contributors=res.contributors, err=err, mask=mask, dt=dt, updates=rep.updates_applied,
                gs=res.global_step)


def test_backup_workers_drop_exactly_the_straggler_and_nobody_waits_for_it():
    world, m, slow = 3, 2, 1
    out = run_workers(_backup_worker, world, f"dtf_t_bk_{os.getpid()}", m, slow)
    assert [o["included"] for o in out] == [True, False, True]
    assert all(o["contributors"] == m for o in out)
    assert all(o["mask"] == 0b101 for o in out)               # first M arrivals, agreed by all replicas
    assert all(o["err"] < 1e-6 for o in out)                  # update = mean of the two fresh gradients
    assert all(o["updates"] == 1 and o["gs"] == 1 for o in out)   # the straggler moved on too
    # the property that matters: the fast replicas' step does not contain the straggler's 1.5 s
    assert out[0]["dt"] < 0.75 and out[2]["dt"] < 0.75, [o["dt"] for o in out]


def _backup_stress_worker(rank, world, board_name, m, steps):
    import random

    from distributed_tensorflow_amd.parallel.strategies import Backup

    comm, board, rep, images, labels = _setup(rank, world, board_name)
    strat = Backup(comm, board, m)
    rng = random.Random(7 + rank)
    included = 0
    it = 0
    while True:
        g = torch.Generator().manual_seed(1000 * it + rank)
        rep.load_batch(torch.randn(BATCH, 32, 32, 3, generator=g), torch.randint(0, 10, (BATCH,), generator=g))
        if rng.random() < 0.3:
            time.sleep(rng.random() * 0.03)                    # uneven load: random per-step delays
        res = strat.step(rep, it, 1e-3)
        included += int(res.included)
        it += 1
        if res.global_step >= steps:
            break
    strat.finish()
    psum = rep.arena.params.double().sum().item()
    comm.close()
    return dict(psum=psum, version=rep.updates_applied, included=included, iters=it)


def test_backup_stress_uneven_load_keeps_replicas_identical():
    world, m, steps = 3, 2, 60
    out = run_workers(_backup_stress_worker, world, f"dtf_t_bks_{os.getpid()}", m, steps, timeout=600)
    assert len({o["version"] for o in out}) == 1 and out[0]["version"] >= steps
    assert len({round(o["psum"], 6) for o in out}) == 1       # same updates in the same order everywhere
    assert sum(o["included"] for o in out) == m * out[0]["version"]   # every update took exactly M gradients


def _soft_kill_worker(rank, world, board_name, k, slow_rank):
    from distributed_tensorflow_amd.parallel.strategies import SoftKill

    comm, board, rep, images, labels = _setup(rank, world, board_name)
    strat = SoftKill(comm, board, k)
    strat._get_log(rep)
    comm.barrier()
    if rank == slow_rank:
        # arrive after the others are done, so the stop flag is already up when backprop starts
        deadline = time.time() + 30
        while board.agg_open_count() < k and time.time() < deadline:
            time.sleep(0.01)
    res = strat.step(rep, 0, 1e-3)
    strat.finish()
    comm.close()
    return dict(aborted=res.aborted, contributors=res.contributors, psum=rep.arena.params.sum().item())


def test_soft_kill_short_circuits_the_straggler():
    world, k, slow = 3, 2, 2
    out = run_workers(_soft_kill_worker, world, f"dtf_t_sk_{os.getpid()}", k, slow)
    assert [o["aborted"] for o in out] == [False, False, True]
    assert all(o["contributors"] == world for o in out)       # zeros are still "pushed" (mean over N)
    assert len({round(o["psum"], 4) for o in out}) == 1


def _interval_worker(rank, world, board_name, interval_ms, slow_rank, run_s):
    from distributed_tensorflow_amd.parallel.strategies import Interval

    comm, board, rep, images, labels = _setup(rank, world, board_name)
    strat = Interval(comm, board, interval_ms)
    strat._get_log(rep)
    comm.barrier()
    t_end = time.time() + run_s
    pushes = accepted = 0
    longest = 0.0
    it = 0
    while time.time() < t_end:
        if rank == slow_rank:
            time.sleep(3 * interval_ms / 1e3)                  # three windows per gradient
        else:
            time.sleep(0.6 * interval_ms / 1e3)
        t0 = time.time()
        res = strat.step(rep, it, 1e-3)
        longest = max(longest, time.time() - t0)
        pushes += 1
        accepted += int(res.included)
        it += 1
    comm.barrier()
    strat.log.catch_up(1e-3, reduce_own=False)
    closed = strat.windows_closed
    version = strat.log.version
    hist = []
    for u in range(max(0, version - 8), version):
        count, mask, owners, scale = board.agg_entry(u)
        hist.append((count, mask))
    strat.finish()
    comm.close()
    return dict(pushes=pushes, accepted=accepted, version=version, closed=closed, psum=rep.arena.params.double().sum().item(),
                longest=longest, hist=hist)


def test_interval_timer_closes_windows_while_workers_never_wait():
    world, slow, interval_ms, run_s = 3, 2, 120, 3.0
    out = run_workers(_interval_worker, world, f"dtf_t_iv_{os.getpid()}", interval_ms, slow, run_s)
    version = out[0]["version"]
    assert len({o["version"] for o in out}) == 1
    assert len({round(o["psum"], 6) for o in out}) == 1
    # global_step advances at the timer's rate, not at the slow replica's
    assert 0.6 * run_s * 1e3 / interval_ms <= version <= 1.3 * run_s * 1e3 / interval_ms, version
    assert out[0]["closed"] == version
    # the slow replica pushes about every third window and every push is accepted (never stale here)
    assert out[slow]["accepted"] == out[slow]["pushes"]
    assert 0.2 * version <= out[slow]["pushes"] <= 0.5 * version, (out[slow]["pushes"], version)
    with_slow = sum(1 for c, m in out[0]["hist"] if (m >> slow) & 1)
    assert 1 <= with_slow <= 5
    # pushing never blocks on the timer or on another replica (a step here is compute + push)
    assert all(o["longest"] < 3 * interval_ms / 1e3 + 1.0 for o in out)


def _weighted_worker(rank, world, board_name):
    from distributed_tensorflow_amd.parallel.strategies import Weighted

    comm, board, rep, images, labels = _setup(rank, world, board_name)
    strat = Weighted(comm, board, num_worker_kill=1, random_seed=142)
    w = strat.weights_for_step(0)
    res = strat.step(rep, 0, 1e-3)
    strat.finish()
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


def _async_ckpt_worker(rank, world, board_name, tmp):
    from distributed_tensorflow_amd.checkpoint.saver import Saver
    from distributed_tensorflow_amd.parallel.async_ps import AsyncParameterShards
    from distributed_tensorflow_amd.train.replica import TorchReplica

    comm, board, rep, images, labels = _setup(rank, world, board_name)
    shards = AsyncParameterShards(rep, comm, board, 8, board_name + "_ps", lambda v: 1e-3)
    for _ in range(4):
        shards.step()
    comm.barrier()
    shards.service()                                           # nothing left in any inbox
    comm.barrier()
    lo, hi = shards.bounds[rank]
    own = {k: shards.slot_regions[rank][k].clone() for k in ("m", "v")}
    own_p = shards.shards[rank].clone()
    versions = shards.versions()
    # without the gather the chief's arena holds no other owner's slots (the defect the gather is for)
    blank_outside = float(rep.arena.full("slot:m")[shards.bounds[1][0]:].abs().max()) if rank == 0 else 0.0
    prefix = None
    if rank == 0:
        shards.gather_state()
        prefix = Saver(rep.arena, "adam").save(tmp, versions[0], versions[0])
    prefix = comm.all_gather_object(prefix)[0]
    shards.close()
    comm.barrier()
    # resume: fresh replicas restore the checkpoint; every owner's shard continues where it was
    rep2 = TorchReplica("resnet", BATCH, num_residual_blocks=1, optimizer="adam")
    rep2.load_batch(images, labels)
    step = Saver(rep2.arena, "adam").restore(prefix)
    shards2 = AsyncParameterShards(rep2, comm, board, 8, board_name + "_ps2", lambda v: 1e-3, start_version=step)
    err = max(float((shards2.slot_regions[rank][k] - own[k]).abs().max()) for k in ("m", "v"))
    err_p = float((shards2.shards[rank] - own_p).abs().max())
    nonzero = float(own["m"].abs().sum())
    v0 = shards2.versions()
    comm.barrier()                      # nobody pushes before everybody has looked
    shards2.step()
    shards2.drain(60)
    v1 = shards2.versions()
    shards2.close()
    comm.barrier()
    comm.close()
    return dict(err=err, err_p=err_p, nonzero=nonzero, versions=versions, step=step, v0=v0, v1=v1, blank_outside=blank_outside)


def test_async_checkpoint_holds_every_owners_optimizer_state_and_resumes(tmp_path):
    world = 2
    out = run_workers(_async_ckpt_worker, world, f"dtf_t_asck_{os.getpid()}", str(tmp_path))
    assert out[0]["blank_outside"] == 0.0
    for o in out:
        assert o["nonzero"] > 0 and o["err"] == 0.0 and o["err_p"] == 0.0      # every owner's m / v / parameters came back
        assert o["step"] == o["versions"][0] == world * 4
        assert o["v0"] == [o["step"]] * world                                  # shard versions continue at the restored step
        assert o["v1"] == [o["step"] + world] * world                          # ... and count on from there


def _sharded_ckpt_worker(rank, world, board_name, tmp):
    from distributed_tensorflow_amd.checkpoint.saver import Saver
    from distributed_tensorflow_amd.parallel.strategies import FullSync

    comm, board, rep, images, labels = _setup(rank, world, board_name)
    strat = FullSync(comm, board, reduce_scatter=True)
    for i in range(3):
        strat.step(rep, i, 1e-3)
    # without the gather the chief's slots are current only inside its own shard
    shard = rep.arena.shard_size(world)
    stale_outside = float(rep.arena.slots["m"][shard:].abs().max()) if rank == 0 else 0.0
    strat.gather_optimizer_state(rep)
    saver = Saver(rep.arena, "adam")
    prefix = None
    if rank == 0:
        prefix = saver.save(tmp, 3, rep.updates_applied)
    prefix = comm.all_gather_object(prefix)[0]
    strat.step(rep, 3, 1e-3)                                   # the uninterrupted run's 4th step
    want = rep.arena.params.clone()
    # resume: fresh replica restored from the checkpoint, same 4th step
    from distributed_tensorflow_amd.train.replica import TorchReplica

    rep2 = TorchReplica("resnet", BATCH, num_residual_blocks=1, optimizer="adam")
    rep2.load_batch(images, labels)
    step = Saver(rep2.arena, "adam").restore(prefix)
    rep2.updates_applied = step
    FullSync(comm, board, reduce_scatter=True).step(rep2, 3, 1e-3)
    err = float((rep2.arena.params - want).abs().max())
    m_nonzero = float(rep2.arena.slots["m"].abs().sum())
    comm.barrier()
    comm.close()
    return dict(err=err, stale_outside=stale_outside, m_nonzero=m_nonzero, step=step)


def test_sharded_optimizer_checkpoint_holds_every_shards_slots(tmp_path):
    out = run_workers(_sharded_ckpt_worker, 3, f"dtf_t_rsck_{os.getpid()}", str(tmp_path))
    assert out[0]["stale_outside"] == 0.0                      # (the defect the gather exists for)
    assert all(o["step"] == 3 and o["m_nonzero"] > 0 for o in out)
    assert all(o["err"] < 1e-7 for o in out)                   # resumed run == uninterrupted run
