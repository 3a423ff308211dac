"""Non-blocking gradient aggregation: a replicated log of reduced gradients.

The reference's straggler policies live on a parameter server: workers push gradients into
ConditionalAccumulators, the chief takes the mean of the first M fresh ones (or of whatever arrived
when its interval timer fires), applies it, and releases ALL workers; a straggler's late gradient
is dropped and the straggler simply reads the newest parameters when it starts its next step
(reference: DistributedLeNet/src/backup_worker_experiment/backup_worker_optimizer.py:241-307,
DistributedResNet/src/sync_replicas_optimizer_modified/sync_replicas_optimizer_modified.py:207-214,402-430).
Nobody ever waits for a worker whose gradient is not part of the update.

Without a parameter server the same behaviour comes from a replicated state machine: every replica
applies the SAME sequence of reduced gradients ("updates") with the same deterministic optimizer,
so parameters stay identical without ever being broadcast.

  * Update u is OPEN while gradients computed on version u arrive. An arrival is one
    compare-exchange on a shared word (update number | count | contributor mask).
  * The arrival that completes the quorum - or the chief's interval timer - CLOSES u: the record
    (contributors, owners of the reduced shards, scale) is published and u+1 opens.
  * Contributors wait only for contributors: each owner sums its shard over the contributors'
    staged gradients and publishes it; then every replica gathers the shards and applies the update.
  * A replica that was not waiting (a straggler, a killed worker, anybody in interval mode) finds
    update u closed when it next looks, drops its stale gradient, and applies u (and every later
    closed update) from the owners' published shards, which stay readable for `ring` updates.
    Replicas that contribute never run more than `ring` updates ahead of the slowest live one.

This module is the host implementation (CPU replicas; control words on the shared-memory board,
data in shared file mappings). The device implementation with the same protocol is
`parallel/update_log_hip.py` (one kernel per step over peer-mapped HBM).
"""
from __future__ import annotations

import os
import threading
import time
from dataclasses import dataclass
from typing import Callable, List, Optional, Tuple, Union

import torch

from .. import _host
from ..runtime.arena import ALIGN_FLOATS

LrLike = Union[float, Callable[[int], float]]


def _shm_tensor(name: str, numel: int, create: bool) -> torch.Tensor:
    path = os.path.join("/dev/shm", name)
    if create:
        with open(path, "wb") as f:
            f.truncate(max(4, numel * 4))
    else:
        deadline = time.time() + 60
        while not os.path.exists(path) or os.path.getsize(path) < numel * 4:
            if time.time() > deadline:
                raise RuntimeError(f"shared region {path} never appeared")
            time.sleep(0.005)
    return torch.from_file(path, shared=True, size=max(1, numel), dtype=torch.float32)


def owner_ranges(total: int, owners: int) -> List[Tuple[int, int, int]]:
    """(owner rank, lo, hi) of every reduced shard: [0, total) cut into equal 64-float-aligned
    chunks, one per set bit of `owners`, in rank order."""
    ranks = [r for r in range(_host.AGG_MAX_RANKS) if (owners >> r) & 1]
    n = len(ranks)
    quanta = (total + ALIGN_FLOATS - 1) // ALIGN_FLOATS
    chunk = (quanta + n - 1) // n * ALIGN_FLOATS
    out = []
    for p, r in enumerate(ranks):
        lo, hi = min(total, p * chunk), min(total, (p + 1) * chunk)
        out.append((r, lo, hi))
    return out


class ShmPlane:
    """Data regions of the host update log in /dev/shm: every replica creates its own staging
    vector and ring of reduced vectors, then maps everybody's."""

    def __init__(self, name: str, rank: int, world: int, total: int, ring: int):
        self._name, self.rank, self.world, self.total, self.ring = name, rank, world, total, ring
        self._files = [f"{name}_stage_{rank}", f"{name}_red_{rank}"]
        _shm_tensor(self._files[0], total, True)
        _shm_tensor(self._files[1], total * ring, True)
        self.stage: List[torch.Tensor] = []
        self.red: List[torch.Tensor] = []

    def map_all(self) -> None:
        self.stage = [_shm_tensor(f"{self._name}_stage_{r}", self.total, False) for r in range(self.world)]
        self.red = [_shm_tensor(f"{self._name}_red_{r}", self.total * self.ring, False).view(self.ring, self.total)
                    for r in range(self.world)]

    def stage_put(self, grads: torch.Tensor, weight: float = 1.0, accumulate: bool = False) -> None:
        st = self.stage[self.rank]
        if accumulate:
            st.add_(grads.detach().to("cpu") if weight == 1.0 else grads.detach().to("cpu") * weight)
        elif weight == 1.0:
            st.copy_(grads)
        elif grads.is_cuda:
            st.copy_(grads * weight)
        else:
            torch.mul(grads, weight, out=st)

    def stage_get(self, q: int, lo: int, hi: int) -> torch.Tensor:
        return self.stage[q][lo:hi]

    def red_put(self, slot: int, lo: int, hi: int, t: torch.Tensor, owner: Optional[int] = None) -> None:
        self.red[self.rank if owner is None else owner][slot][lo:hi].copy_(t)

    def red_get(self, r: int, slot: int, lo: int, hi: int) -> torch.Tensor:
        return self.red[r][slot][lo:hi]

    def close(self) -> None:
        for f in self._files:
            try:
                os.unlink(os.path.join("/dev/shm", f))
            except OSError:
                pass
        self._files = []


def lr_at(lr: LrLike, update: int) -> float:
    return float(lr(update)) if callable(lr) else float(lr)


@dataclass
class ContributionResult:
    accepted: bool        # the gradient entered an update (False: it was stale or late and was dropped)
    order: int            # gradients accepted before it
    applied: int          # updates this replica applied during the call
    version: int          # this replica's version after the call


class HostUpdateLog:
    """Update log over host memory for `TorchReplica`s of one node."""

    def __init__(self, replica, comm, board, name: str, timeout_s: float = 600.0, start_version: int = 0):
        if board is None:
            raise ValueError("the update log needs the shared-memory control board")
        if comm.world > _host.AGG_MAX_RANKS:
            raise ValueError(f"the update log serves at most {_host.AGG_MAX_RANKS} replicas")
        self.replica, self.comm, self.board = replica, comm, board
        self.rank, self.world = comm.rank, comm.world
        self.ring = _host.AGG_RING
        self.total = replica.arena.total
        self.timeout_s = timeout_s
        self.version = start_version
        self._name = name.strip("/").replace("/", "_")
        self._lock = threading.Lock()          # serialises the training thread and the chief's timer thread
        self._pending: Optional[Tuple[int, int]] = None   # interval mode: (update, gradients) staged and not yet known closed
        self.dropped = 0
        self.contributed = 0
        self.max_version = 1 << 60
        self._last_lr: LrLike = 0.0
        # data regions: shared memory on one node, values of the job's key-value store across machines
        from .store_board import StoreBoard, StorePlane

        if isinstance(board, StoreBoard):
            self.plane = StorePlane(board, self._name, self.rank, self.world, self.total, self.ring)
        else:
            self.plane = ShmPlane(self._name, self.rank, self.world, self.total, self.ring)
        if comm.is_chief:
            board.agg_reset(start_version)
        comm.barrier()
        if isinstance(self.plane, ShmPlane):
            self.plane.map_all()
        comm.barrier()

    # ---------------------------------------------------------------------------------------------
    def forward_backward(self, replica, abortable: bool = True) -> bool:
        """Runs the replica's forward + backward pass; with `abortable` the backward pass polls the
        abort word at every layer and stops (zeros for the layers not reached) once the update it
        works for is complete. Returns False when it was cut short."""
        done = replica.forward_backward(should_abort=self.abort_requested if abortable else None)
        # a GPU replica only enqueues the pass on its own stream; this log stages the gradient with
        # host-side copies, which must not start before the pass has finished (no-op on CPU replicas)
        replica.synchronize()
        return done

    def abort_requested(self) -> bool:
        """True once the update this replica is computing a gradient for has all it will take."""
        return self.board.agg_abort() > self.version

    def global_version(self) -> int:
        return int(self.board.agg_open_update())

    def sync_version(self) -> int:
        return self.version

    def contributors_hint(self) -> int:
        """Gradients in the window that is open now."""
        return int(self.board.agg_open_count())

    def _check(self, ok: bool, what: str) -> None:
        if not ok:
            raise TimeoutError(f"update log (rank {self.rank}, version {self.version}): timed out waiting for {what}")

    def _throttle(self, update: int) -> None:
        # ring slot of `update` was last used by update - ring: every live replica must have applied that
        self._check(self.board.agg_wait_min_applied(update - self.ring + 1, self.timeout_s),
                    f"the slowest replica to reach version {update - self.ring + 1}")

    # ---------------------------------------------------------------------------------------------
    def contribute(self, lr: LrLike, *, weight: float = 1.0, quorum: int, abort_at: int = 0, scale: float = 0.0) -> ContributionResult:
        """Offers this replica's gradient (arena.grads) for the update that turns its version into the
        next one; waits for that update to close only if the gradient was accepted; then applies
        every closed update it has not applied yet."""
        a = self.replica.arena
        e = self.version
        if e >= self.max_version:                     # the run is over: nothing is offered any more
            return ContributionResult(False, -1, self.catch_up(lr), self.version)
        self._throttle(e)
        self.plane.stage_put(a.grads, weight)
        status, order, _closed, _open_u = self.board.agg_arrive(e, self.rank, 1, quorum, abort_at, scale, False)
        accepted = status == 0
        if accepted:
            self.contributed += 1
            self._check(self.board.agg_wait_closed(e, self.timeout_s), f"update {e} to collect its {quorum} gradients")
        else:
            self.dropped += 1
        applied = self.catch_up(lr)
        return ContributionResult(accepted, order, applied, self.version)

    def skip(self, lr: LrLike) -> ContributionResult:
        """A replica that contributes nothing to the current update (a "killed" worker of the
        weighted scheme): waits for the others to close it, then applies it."""
        e = self.version
        self._check(self.board.agg_wait_closed(e, self.timeout_s), f"update {e} to close")
        return ContributionResult(False, -1, self.catch_up(lr), self.version)

    def contribute_open(self, lr: LrLike, max_staleness: int = -1) -> ContributionResult:
        """Interval mode: adds the gradient to whatever update is open and returns at once (never
        waits for the window to close); several gradients of one replica within a window are summed
        locally and counted separately in the divisor."""
        a = self.replica.arena
        b = self.board
        with self._lock:
            count = 1
            if self._pending is not None:
                pu, pc = self._pending
                if b.agg_take_back(pu, self.rank, pc):
                    self.plane.stage_put(a.grads, accumulate=True)   # window still open: nobody reads the staged sum
                    count = pc + 1
                else:
                    # the window closed with the staged sum in it: the chief may still be reading
                    self._check(b.agg_wait_closed(pu, self.timeout_s), f"update {pu} to close")
                    ent = b.agg_entry(pu)
                    self._check(b.agg_wait_ready(pu, ent[2], self.timeout_s), f"update {pu} to be reduced")
                    self.plane.stage_put(a.grads)
            else:
                self.plane.stage_put(a.grads)
            open_u = b.agg_open_update()
            if (max_staleness >= 0 and open_u - self.version > max_staleness) or open_u >= self.max_version:
                self._pending = None
                self.dropped += 1
                accepted, order = False, -1
            else:
                self._throttle(open_u)
                status, order, _c, open_u = b.agg_arrive(self.version, self.rank, count, 0, 0, 0.0, True)
                self._pending = (open_u, count)
                self.contributed += 1
                accepted = True
        applied = self.catch_up(lr, reduce_own=False)   # windows are reduced by the chief's timer
        return ContributionResult(accepted, order, applied, self.version)

    def close_window(self) -> int:
        """Chief's interval timer: closes the open update if anything arrived, reduces it (the
        chief owns the whole vector) and publishes it. Returns the closed update or -1."""
        b = self.board
        u = b.agg_open_update()
        if b.agg_min_applied() < u - self.ring + 1:
            return -1                              # a live replica still needs the slot: skip this tick
        u = b.agg_close(self.rank)
        if u < 0:
            return -1
        count, mask, owners, _scale = b.agg_entry(u)
        self.plane.red_put(u % self.ring, 0, self.total, self._sum_staged(mask, 0, self.total))
        b.agg_publish_ready(u, self.rank)
        return u

    def _sum_staged(self, mask: int, lo: int, hi: int) -> torch.Tensor:
        """Sum over the contributors in `mask` of their staged gradients' [lo, hi)."""
        acc = None
        for q in range(self.world):
            if (mask >> q) & 1:
                part = self.plane.stage_get(q, lo, hi)
                acc = part.clone() if acc is None else acc.add_(part)
        return acc if acc is not None else torch.zeros(hi - lo)

    # ---------------------------------------------------------------------------------------------
    def catch_up(self, lr: LrLike, reduce_own: bool = True) -> int:
        """Applies every closed update this replica has not applied yet; returns how many."""
        b = self.board
        a = self.replica.arena
        n = 0
        self._last_lr = lr
        while True:
            u = self.version
            ent = b.agg_entry(u)
            if ent is None:
                return n
            count, mask, owners, scale = ent
            slot = u % self.ring
            shards = owner_ranges(self.total, owners)
            if reduce_own and (owners >> self.rank) & 1 and owners == mask and not b.agg_is_ready(u, self.rank):
                # owners == mask: contributors reduce their own shards (two-shot); an update closed by
                # the chief's timer is reduced by the timer thread instead
                for r, lo, hi in shards:
                    if r != self.rank or hi <= lo:
                        continue
                    self.plane.red_put(slot, lo, hi, self._sum_staged(mask, lo, hi))
                b.agg_publish_ready(u, self.rank)
            self._wait_shards(u, mask, owners, shards)
            g = a.grads
            for r, lo, hi in shards:
                if hi > lo:
                    g[lo:hi].copy_(self.plane.red_get(r, slot, lo, hi))
            self.replica.updates_applied = u
            self.replica.apply_update(lr_at(lr, u), grad_scale=scale)
            self.version = u + 1
            b.agg_set_applied(self.rank, self.version)
            n += 1

    def _wait_shards(self, u: int, mask: int, owners: int, shards) -> None:
        """Waits for the owners' reduced shards of update u. An owner the watchdog has declared dead
        will never publish its shard: the lowest-ranked live contributor sums it in its place from the
        staged gradients (a dead worker's staged gradient outlives it) into the dead owner's region
        and publishes it in the dead owner's name."""
        b = self.board
        deadline = time.time() + self.timeout_s
        while True:
            if b.agg_wait_ready(u, owners, min(0.5, max(0.01, deadline - time.time()))):
                return
            dead = int(b.agg_dead())
            live = [q for q in range(self.world) if (mask >> q) & 1 and not (dead >> q) & 1]
            if owners & dead and live and live[0] == self.rank:
                for r, lo, hi in shards:
                    if (dead >> r) & 1 and not b.agg_is_ready(u, r):
                        if hi > lo:
                            self.plane.red_put(u % self.ring, lo, hi, self._sum_staged(mask, lo, hi), owner=r)
                        b.agg_publish_ready(u, r)
            self._check(time.time() < deadline and not b.should_stop(), f"the reduced shards of update {u}")

    def finish(self, final_version: Optional[int] = None, collective: bool = True) -> None:
        """End of training: a replica may stop contributing while others still need its published
        shards; keep the mappings until every live replica has applied everything closed."""
        b = self.board
        deadline = time.time() + self.timeout_s
        while time.time() < deadline and not b.should_stop():
            target = b.agg_open_update() if final_version is None else final_version
            if b.agg_min_applied() >= target:
                break
            with self._lock:
                self.catch_up(self._last_lr)   # this replica may be the one the others are waiting for
            time.sleep(0.001)

    def close(self) -> None:
        self.plane.close()


def make_update_log(replica, comm, board, start_version: int = 0, timeout_s: float = 600.0):
    """The update log that matches the replica: host memory for CPU replicas, peer-mapped HBM and
    one kernel per step for GPU replicas."""
    name = "ulog_" + board.name.strip("/")
    from .store_board import StoreBoard

    if isinstance(board, StoreBoard):
        # workers on several machines: gradients are staged through the host and the store, whatever
        # device computed them
        return HostUpdateLog(replica, comm, board, name, timeout_s=timeout_s, start_version=start_version)
    if getattr(replica, "device", None) is not None and replica.device.type == "cuda":
        from .update_log_hip import DeviceLogUnavailable, DeviceUpdateLog

        try:
            return DeviceUpdateLog(replica, comm, board, name, timeout_s=timeout_s, start_version=start_version)
        except DeviceLogUnavailable as e:
            # raised on all ranks together: the host log carries the same protocol through the board
            import logging

            logging.getLogger("dtf").warning("device update log unavailable (%s): using the host update log", e)
    return HostUpdateLog(replica, comm, board, name, timeout_s=timeout_s, start_version=start_version)
