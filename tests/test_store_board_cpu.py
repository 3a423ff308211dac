"""The control board over the process group's key-value store (workers on several machines) must
behave like the shared-memory board: the same random sequence of calls is applied to both."""
import datetime
import os
import random

import pytest
import torch
import torch.distributed as dist

from distributed_tensorflow_amd import _host
from distributed_tensorflow_amd.parallel.store_board import StoreBoard, StorePlane


@pytest.fixture()
def boards():
    store = dist.TCPStore("127.0.0.1", 0, 1, True, timeout=datetime.timedelta(seconds=20))
    name = f"dtf_t_sb_{os.getpid()}_{random.getrandbits(30)}"
    shm = _host.Board(name, 4, True, 0)
    yield shm, StoreBoard(store, name, 4, 0)
    _host.Board.unlink(name)


def test_update_log_words_match_the_shared_memory_board(boards):
    shm, sb = boards
    rng = random.Random(7)
    for b in boards:
        b.agg_reset(3)
    pending = {}
    for step in range(600):
        op = rng.choice(["arrive", "arrive", "arrive", "arrive_any", "take_back", "close", "ready", "applied", "abort", "dead"])
        rank = rng.randrange(4)
        e = shm.agg_open_update() - rng.choice([0, 0, 0, 1])
        if op == "arrive":
            args = (e, rank, 1, rng.choice([0, 2, 3]), rng.choice([0, 2]), rng.choice([0.0, 0.25]), False)
            got = [tuple(b.agg_arrive(*args)) for b in boards]
        elif op == "arrive_any":
            args = (e, rank, rng.choice([1, 2]), 0, 0, 0.0, True)
            got = [tuple(b.agg_arrive(*args)) for b in boards]
            if got[0][0] == 0:
                pending[rank] = (got[0][3], args[2])
        elif op == "take_back":
            u, c = pending.pop(rank, (e, 1))
            got = [b.agg_take_back(u, rank, c) for b in boards]
        elif op == "close":
            got = [b.agg_close(rank) for b in boards]
        elif op == "ready":
            u = max(3, shm.agg_open_update() - 1)
            for b in boards:
                b.agg_publish_ready(u, rank)
            got = [(b.agg_is_ready(u, rank), b.agg_is_ready(u, (rank + 1) % 4), b.agg_wait_ready(u, 1 << rank, 0.2)) for b in boards]
        elif op == "applied":
            v = shm.agg_open_update() - rng.choice([0, 1])
            for b in boards:
                b.agg_set_applied(rank, v)
            got = [(b.agg_applied(rank), b.agg_min_applied()) for b in boards]
        elif op == "abort":
            v = rng.randrange(0, 40)
            for b in boards:
                b.agg_raise_abort(v)
            got = [b.agg_abort() for b in boards]
        else:
            if rng.random() < 0.1:
                for b in boards:
                    b.agg_mark_dead(rank)
            got = [b.agg_dead() for b in boards]
        assert got[0] == got[1], (step, op, got)
        state = [(b.agg_open_update(), b.agg_open_count(), b.agg_open_mask()) for b in boards]
        assert state[0] == state[1], (step, op, state)
        u = shm.agg_open_update() - 1
        if u >= 3:
            ents = [b.agg_entry(u) for b in boards]
            assert (ents[0] is None) == (ents[1] is None)
            if ents[0] is not None:
                assert tuple(ents[0][:3]) == tuple(ents[1][:3]) and abs(ents[0][3] - ents[1][3]) < 1e-6
                assert all(b.agg_wait_closed(u, 0.2) for b in boards)
    assert not sb.agg_wait_closed(shm.agg_open_update() + 5, 0.3)      # a wait that times out says so


def test_driver_side_calls_match(boards):
    shm, sb = boards
    for b in boards:
        b.set_global_step(11)
        b.put_tokens(2, 2, 40)
        b.set_local_step(1, 9)
        b.request_save(17)
        b.post_times(3, 5, 100, 250)
        b.mark_ready(0)
    for b in boards:
        assert b.global_step() == 11 and b.local_step(1) == 9 and b.save_step() == 17
        assert b.take_token(2, 0.5) == 40 and b.take_token(2, 0.5) == 40 and b.take_token(2, 0.05) is None
        assert tuple(b.get_times(3, 5)) == (100, 250) and b.get_times(3, 6) is None
        assert not b.all_ready() and not b.wait_all_ready(0.1)
        for r in range(1, 4):
            b.mark_ready(r)
        assert b.all_ready() and b.wait_all_ready(0.1)
        assert b.last_heartbeat_ns(1) == 0
        b.heartbeat(1)
        assert 0 < b.last_heartbeat_ns(1) < type(b).DEPARTED
        b.depart(1)
        assert b.last_heartbeat_ns(1) == type(b).DEPARTED
        assert not b.should_stop()
        b.request_stop()
        assert b.should_stop()
    with pytest.raises(NotImplementedError):
        sb.ring_doorbell(0, 1, 3)


def test_regions_travel_in_pieces(monkeypatch):
    store = dist.TCPStore("127.0.0.1", 0, 1, True, timeout=datetime.timedelta(seconds=20))
    monkeypatch.setattr(StorePlane, "_PIECE", 1000)
    total, ring = 4321, 4
    planes = [StorePlane(StoreBoard(store, "p", 2, r), "log", r, 2, total, ring) for r in range(2)]
    g = [torch.randn(total), torch.randn(total)]
    planes[0].stage_put(g[0])
    planes[1].stage_put(g[1], weight=0.5)
    planes[1].stage_put(g[1], accumulate=True)                 # interval mode: a second gradient in the same window
    assert torch.equal(planes[1].stage_get(0, 100, 2900), g[0][100:2900])
    assert torch.allclose(planes[0].stage_get(1, 0, total), 1.5 * g[1])
    planes[0].red_put(3, 64, 2200, g[0][64:2200])
    assert torch.equal(planes[1].red_get(0, 3, 64, 2200), g[0][64:2200])
    with pytest.raises(RuntimeError):
        planes[1].red_get(0, 3, 64, 2201)                      # a reader that expects another shard length is told
    for p in planes:
        p.close()


@pytest.mark.parametrize("case", ["backup", "backup_stress", "soft_kill", "interval", "weighted"])
def test_aggregation_policies_over_the_store(case, monkeypatch):
    """The policy tests of test_strategies_cpu.py, with the workers coordinating through the process
    group's store instead of shared memory (DTF_CONTROL_PLANE=store reaches the worker processes)."""
    import test_strategies_cpu as t

    monkeypatch.setenv("DTF_CONTROL_PLANE", "store")
    if case == "interval":
        # same scenario, wider timing bounds: every board call is a network round trip here
        from mp_util import run_workers

        world, slow, interval_ms, run_s = 3, 2, 150, 3.0
        out = run_workers(t._interval_worker, world, f"dtf_t_ivs_{os.getpid()}", interval_ms, slow, run_s)
        version = out[0]["version"]
        assert len({o["version"] for o in out}) == 1 and len({round(o["psum"], 6) for o in out}) == 1
        assert 3 <= version <= 1.3 * run_s * 1e3 / interval_ms, version    # the timer never runs ahead of its period
        assert out[0]["closed"] == version
        assert out[slow]["accepted"] == out[slow]["pushes"] >= 2
        assert any((m >> slow) & 1 for c, m in out[0]["hist"])
        return
    {"backup": t.test_backup_workers_drop_exactly_the_straggler_and_nobody_waits_for_it,
     "backup_stress": t.test_backup_stress_uneven_load_keeps_replicas_identical,
     "soft_kill": t.test_soft_kill_short_circuits_the_straggler,
     "weighted": t.test_weighted_least_squares_aggregation}[case]()
