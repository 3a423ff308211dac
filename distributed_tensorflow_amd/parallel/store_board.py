"""The control board and the host update log's data regions over the process group's key-value store,
for jobs whose workers sit on several machines.

On one node every worker maps the chief's shared-memory `Board` (csrc/host/board.cc) and the
update log's buffers in /dev/shm. The reference's jobs are clusters of machines whose workers talk to a
parameter-server process (reference: DistributedLeNet/src/mnist_distributed_train.py:25-35,
backup_worker_optimizer.py:241-307). For that shape of job the same protocol runs over the store
that torch.distributed already keeps for the job's rendezvous (a TCP store hosted by rank 0; nothing
else listens anywhere): `StoreBoard` offers the part of the `Board` interface that the driver, the
watchdog and `HostUpdateLog` use, built from the store's atomic add / compare-and-set / blocking
wait, and `StorePlane` keeps the staged gradients and the reduced shards as values of that store.

Every call is a network round trip and every gradient crosses the chief's network interface, as
every variable crosses the parameter server's in the reference: this is the functional path for
clusters, not the fast path (one node: shared memory or peer-mapped HBM).

Not offered over the store: the asynchronous shards' doorbells (they move data through mapped
memory) - the driver refuses that combination.
"""
from __future__ import annotations

import datetime
import json
import time
from typing import Optional, Sequence

import torch

from .. import _host

_FROZEN = 1 << 31


def _make_open(u: int, count: int, mask: int, frozen: bool) -> int:
    return (u << 32) | (_FROZEN if frozen else 0) | ((count & 0x7FFF) << 16) | (mask & 0xFFFF)


def _open_u(w: int) -> int:
    return w >> 32


def _open_count(w: int) -> int:
    return (w >> 16) & 0x7FFF


def _open_mask(w: int) -> int:
    return w & 0xFFFF


class StoreBoard:
    """`Board` over a torch.distributed store. `agg_reset` (chief) must run before the update log's
    words are used; the other keys are created on first use."""

    DEPARTED = _host.Board.DEPARTED

    def __init__(self, store, name: str, world: int, rank: int):
        if world > _host.AGG_MAX_RANKS:
            raise ValueError(f"the store-backed board serves at most {_host.AGG_MAX_RANKS} workers")
        self.store = store
        self.name = name
        self.world = world
        self.rank = rank
        self._p = "dtfb/" + name.strip("/").replace("/", "_") + "/"
        self._barriers = 0

    # -- helpers ----------------------------------------------------------------------------------
    def _k(self, key: str) -> str:
        return self._p + key

    def _get_int(self, key: str, default: int = 0) -> int:
        k = self._k(key)
        if default == 0:
            return int(self.store.add(k, 0))     # one round trip; a missing key reads as 0
        if not self.store.check([k]):
            return default
        return int(self.store.get(k).decode())

    def _set_int(self, key: str, value: int) -> None:
        self.store.set(self._k(key), str(int(value)))

    def _cas(self, key: str, old: int, new: int) -> int:
        """Value of `key` after trying to replace `old` by `new` (== new: it was replaced)."""
        return int(self.store.compare_set(self._k(key), str(int(old)), str(int(new))).decode())

    def _wait_keys(self, keys: Sequence[str], timeout_s: float) -> bool:
        """Blocks until all keys exist; gives up at the timeout or when the job's stop flag rises."""
        full = [self._k(k) for k in keys]
        deadline = time.time() + timeout_s
        while True:
            left = deadline - time.time()
            try:
                self.store.wait(full, datetime.timedelta(seconds=max(0.05, min(1.0, left))))
                return True
            except Exception:   # noqa: BLE001 - the store signals a timeout with a generic error
                if self.store.check(full):
                    return True
                if time.time() >= deadline or self.should_stop():
                    return False

    @staticmethod
    def now_ns() -> int:
        return time.time_ns()

    @staticmethod
    def unlink(name: str) -> None:   # nothing outlives the store
        pass

    # -- startup barrier, global step, stop --------------------------------------------------------
    def mark_ready(self, rank: int) -> None:
        self.store.set(self._k(f"ready/{rank}"), "1")

    def all_ready(self) -> bool:
        return bool(self.store.check([self._k(f"ready/{r}") for r in range(self.world)]))

    def wait_all_ready(self, timeout_s: float) -> bool:
        return self._wait_keys([f"ready/{r}" for r in range(self.world)], timeout_s)

    def barrier(self, timeout_s: float = 600.0) -> bool:
        gen = self._barriers
        self._barriers += 1
        if self.store.add(self._k(f"bar/{gen}"), 1) == self.world:
            self.store.set(self._k(f"bar/{gen}/open"), "1")
        return self._wait_keys([f"bar/{gen}/open"], timeout_s)

    def global_step(self) -> int:
        return self._get_int("gstep")

    def set_global_step(self, s: int) -> None:
        self._set_int("gstep", s)

    def advance_global_step(self) -> int:
        return int(self.store.add(self._k("gstep"), 1))

    def request_stop(self) -> None:
        self.store.set(self._k("stop"), "1")

    def should_stop(self) -> bool:
        return bool(self.store.check([self._k("stop")]))

    def report_failure(self, rank: int) -> None:
        self._set_int("failed", rank + 1)
        self.request_stop()

    def failed_rank(self) -> int:
        return self._get_int("failed") - 1

    def set_local_step(self, rank: int, s: int) -> None:
        self._set_int(f"lstep/{rank}", s)

    def local_step(self, rank: int) -> int:
        return self._get_int(f"lstep/{rank}")

    def request_save(self, iteration: int) -> None:
        self._set_int("save_step", iteration)

    def save_step(self) -> int:
        return self._get_int("save_step", -1)

    # -- tokens (reference: sync_token_q) -----------------------------------------------------------
    def put_tokens(self, rank: int, n: int, step: int) -> None:
        self._set_int(f"tokstep/{rank}", step)
        self.store.add(self._k(f"tok/{rank}"), int(n))

    def take_token(self, rank: int, timeout_s: float) -> Optional[int]:
        deadline = time.time() + timeout_s
        k = self._k(f"tok/{rank}")
        while True:
            if self.store.add(k, 0) > 0:          # only `rank` itself takes its tokens
                self.store.add(k, -1)
                return self._get_int(f"tokstep/{rank}", -1)
            if time.time() > deadline or self.should_stop():
                return None
            time.sleep(0.002)

    def token_depth(self, rank: int) -> int:
        return int(self.store.add(self._k(f"tok/{rank}"), 0))

    # -- heartbeats (every worker stamps with its own clock: keep the machines' clocks within a
    # fraction of the silence threshold of each other) ------------------------------------------------
    def heartbeat(self, rank: int) -> None:
        self._set_int(f"hb/{rank}", time.time_ns())

    def last_heartbeat_ns(self, rank: int) -> int:
        return self._get_int(f"hb/{rank}")

    def depart(self, rank: int) -> None:
        self._set_int(f"hb/{rank}", self.DEPARTED)

    # -- timing telemetry ---------------------------------------------------------------------------
    def post_times(self, rank: int, iteration: int, t_dequeue_ns: int, t_finish_ns: int) -> None:
        self.store.set(self._k(f"times/{rank}/{iteration % _host.BOARD_RING}"), json.dumps([iteration, t_dequeue_ns, t_finish_ns]))

    def get_times(self, rank: int, iteration: int):
        k = self._k(f"times/{rank}/{iteration % _host.BOARD_RING}")
        if not self.store.check([k]):
            return None
        it, a, b = json.loads(self.store.get(k).decode())
        return (a, b) if it == iteration else None

    # -- asynchronous shards: data moves through mapped memory, which a store cannot offer --------
    def _no_async(self, *_a):
        raise NotImplementedError("asynchronous parameter shards need workers that can map each other's memory (one node)")

    shard_version = bump_shard_version = set_shard_version = ring_doorbell = poll_doorbell = clear_doorbell = _no_async

    # -- update log (semantics of csrc/host/board.cc, word for word) --------------------------------
    def agg_reset(self, u0: int) -> None:
        self._set_int("abort", 0)
        self._set_int("dead", 0)
        for r in range(self.world):
            self._set_int(f"applied/{r}", u0)
        self._set_int("open", _make_open(u0, 0, 0, False))

    def _open(self) -> int:
        return int(self.store.get(self._k("open")).decode())

    def _publish_closed(self, u: int, count: int, mask: int, owners: int, scale: float) -> None:
        self.store.set(self._k(f"hist/{u}"), json.dumps([count, mask, owners, scale]))
        old = u - 4 * _host.AGG_RING          # records nobody can need any more
        if old >= 0:
            self.store.delete_key(self._k(f"hist/{old}"))
            for r in range(self.world):
                self.store.delete_key(self._k(f"rdy/{old}/{r}"))
        self._set_int("open", _make_open(u + 1, 0, 0, False))

    def agg_arrive(self, e: int, rank: int, add: int = 1, quorum: int = 0, abort_at: int = 0, scale: float = 0.0,
                   any_update: bool = False):
        w = self._open()
        spins = 0
        while True:
            u = _open_u(w)
            if not any_update and u != e:
                return 1, 0, False, u          # another update is open: the gradient is stale
            if w & _FROZEN:
                if not any_update:
                    return 2, 0, False, u      # quorum complete, the closer is publishing: too late
                spins += 1
                time.sleep(min(0.002, 0.0001 * spins))
                w = self._open()
                continue
            c = _open_count(w)
            c2 = c + add
            m2 = _open_mask(w) | (1 << rank)
            closes = quorum > 0 and c2 >= quorum
            new = _make_open(u, c2, m2, closes)
            cur = self._cas("open", w, new)
            if cur == new:
                if abort_at > 0 and c2 >= abort_at:
                    self.agg_raise_abort(u + 1)
                if closes:
                    self._publish_closed(u, c2, m2, m2, scale if scale > 0 else 1.0 / c2)
                    return 0, c, True, u + 1
                return 0, c, False, u
            w = cur

    def agg_take_back(self, e: int, rank: int, count: int) -> bool:
        w = self._open()
        while True:
            if _open_u(w) != e or (w & _FROZEN) or not (_open_mask(w) & (1 << rank)) or _open_count(w) < count:
                return False
            new = _make_open(e, _open_count(w) - count, _open_mask(w) & ~(1 << rank), False)
            cur = self._cas("open", w, new)
            if cur == new:
                return True
            w = cur

    def agg_close(self, owner_rank: int) -> int:
        w = self._open()
        while True:
            if (w & _FROZEN) or _open_count(w) < 1:
                return -1
            cur = self._cas("open", w, w | _FROZEN)
            if cur == (w | _FROZEN):
                break
            w = cur
        u = _open_u(w)
        self._publish_closed(u, _open_count(w), _open_mask(w), 1 << owner_rank, 1.0 / _open_count(w))
        return u

    def agg_open_update(self) -> int:
        return _open_u(self._open())

    def agg_open_count(self) -> int:
        return _open_count(self._open())

    def agg_open_mask(self) -> int:
        return _open_mask(self._open())

    def agg_entry(self, u: int):
        k = self._k(f"hist/{u}")
        if not self.store.check([k]):
            if u + 4 * _host.AGG_RING < self.agg_open_update():
                raise RuntimeError(f"update log: update {u} was recycled (replica lags more than the ring)")
            return None
        count, mask, owners, scale = json.loads(self.store.get(k).decode())
        return count, mask, owners, scale

    def agg_wait_closed(self, u: int, timeout_s: float) -> bool:
        return self._wait_keys([f"hist/{u}"], timeout_s)

    def agg_publish_ready(self, u: int, owner: int) -> None:
        self.store.set(self._k(f"rdy/{u}/{owner}"), "1")

    def agg_is_ready(self, u: int, owner: int) -> bool:
        return bool(self.store.check([self._k(f"rdy/{u}/{owner}")]))

    def agg_wait_ready(self, u: int, owners: int, timeout_s: float) -> bool:
        return self._wait_keys([f"rdy/{u}/{p}" for p in range(_host.AGG_MAX_RANKS) if (owners >> p) & 1], timeout_s)

    def agg_set_applied(self, rank: int, version: int) -> None:
        self._set_int(f"applied/{rank}", version)

    def agg_applied(self, rank: int) -> int:
        return self._get_int(f"applied/{rank}")

    def agg_min_applied(self) -> int:
        dead = self.agg_dead()
        live = [r for r in range(self.world) if not (dead >> r) & 1]
        if not live:
            return 1 << 62
        vals = self.store.multi_get([self._k(f"applied/{r}") for r in live])
        return min(int(v.decode()) for v in vals)

    def agg_wait_min_applied(self, version: int, timeout_s: float) -> bool:
        deadline = time.time() + timeout_s
        spins = 0
        while self.agg_min_applied() < version:
            if time.time() > deadline or self.should_stop():
                return False
            spins += 1
            time.sleep(min(0.005, 0.0002 * spins))
        return True

    def agg_abort(self) -> int:
        return self._get_int("abort")

    def agg_raise_abort(self, value: int) -> None:
        cur = self._get_int("abort")
        while cur < value:
            got = self._cas("abort", cur, value)
            if got == value:
                return
            cur = got

    def agg_mark_dead(self, rank: int) -> None:
        cur = self._get_int("dead")
        while not (cur >> rank) & 1:
            got = self._cas("dead", cur, cur | (1 << rank))
            if got == (cur | (1 << rank)):
                return
            cur = got

    def agg_dead(self) -> int:
        return self._get_int("dead")


class StorePlane:
    """Data regions of `HostUpdateLog` as store values: `stage/<rank>` holds a worker's staged
    gradient (whole vector), `red/<slot>/<owner>` an owner's reduced shard of the update in that ring
    slot. The update log's protocol orders the accesses (a region is rewritten only after everybody
    who reads it has said so on the board), exactly as for the shared-memory regions."""

    def __init__(self, board: StoreBoard, name: str, rank: int, world: int, total: int, ring: int):
        self.store = board.store
        self._p = board._p + "data/" + name + "/"
        self.rank, self.world, self.total, self.ring = rank, world, total, ring
        self._staged: Optional[torch.Tensor] = None

    # the store refuses values above 8 MiB: a region travels in pieces of at most _PIECE floats, whose
    # number follows from the region's length, which both sides know
    _PIECE = 1 << 20

    def _put(self, key: str, t: torch.Tensor) -> None:
        flat = t.detach().to("cpu", torch.float32).contiguous().view(-1)
        data = flat.numpy()
        for i, lo in enumerate(range(0, max(1, flat.numel()), self._PIECE)):
            self.store.set(self._p + f"{key}/{i}", data[lo:lo + self._PIECE].tobytes())

    def _fetch(self, key: str, numel: int, lo: int = 0, hi: Optional[int] = None) -> torch.Tensor:
        """Values [lo, hi) of a region of `numel` values: only the pieces that overlap travel."""
        hi = numel if hi is None else hi
        if hi <= lo:
            return torch.empty(0)
        first, last = lo // self._PIECE, (hi - 1) // self._PIECE
        data = b"".join(self.store.multi_get([self._p + f"{key}/{i}" for i in range(first, last + 1)]))
        want = min(numel, (last + 1) * self._PIECE) - first * self._PIECE
        if len(data) != want * 4:
            raise RuntimeError(f"update log: region {key} holds {len(data) // 4} values where {want} were expected")
        t = torch.frombuffer(bytearray(data), dtype=torch.float32)
        return t[lo - first * self._PIECE:hi - first * self._PIECE]

    # staged gradient of this worker
    def stage_put(self, grads: torch.Tensor, weight: float = 1.0, accumulate: bool = False) -> None:
        g = grads.detach().to("cpu", torch.float32)
        if weight != 1.0:
            g = g * weight
        if accumulate and self._staged is not None:
            g = self._staged + g
        self._staged = g.clone() if g.data_ptr() == grads.data_ptr() else g
        self._put(f"stage/{self.rank}", self._staged)

    def stage_get(self, q: int, lo: int, hi: int) -> torch.Tensor:
        return self._fetch(f"stage/{q}", self.total, lo, hi)

    # reduced shards
    def red_put(self, slot: int, lo: int, hi: int, t: torch.Tensor, owner: Optional[int] = None) -> None:
        if t.numel() != hi - lo:
            raise ValueError("red_put: shard length mismatch")
        self._put(f"red/{slot}/{self.rank if owner is None else owner}", t)

    def red_get(self, r: int, slot: int, lo: int, hi: int) -> torch.Tensor:
        return self._fetch(f"red/{slot}/{r}", hi - lo)

    def close(self) -> None:
        pieces = max(1, -(-self.total // self._PIECE))
        for key in [f"stage/{self.rank}"] + [f"red/{s}/{self.rank}" for s in range(self.ring)]:
            for i in range(pieces):
                try:
                    self.store.delete_key(self._p + f"{key}/{i}")
                except Exception:   # noqa: BLE001 - the store may be gone at shutdown
                    return


def default_store():
    """The key-value store of the default process group (hosted by rank 0 for the rendezvous)."""
    import torch.distributed as dist
    from torch.distributed import distributed_c10d

    if not dist.is_initialized():
        raise RuntimeError("the store-backed control board needs an initialised process group")
    return distributed_c10d._get_default_store()
