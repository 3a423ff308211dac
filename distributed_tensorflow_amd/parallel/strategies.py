"""Gradient-aggregation policies: what the reference implements as TensorFlow optimizer wrappers
around parameter-server accumulators and token queues, re-expressed without a parameter server.

Every policy keeps ALL replicas' parameters identical: each replica runs the same fused optimizer
on the same aggregated gradients in the same order, so no parameter broadcast is needed.

  FullSync      tf.train.SyncReplicasOptimizer(V2): mean of all N gradients
                (reference: DistributedLeNet/src/distributed_train.py:181-185): bucketed all-reduce
                overlapped with backward, or reduce-scatter + per-shard optimizer + all-gather.
The straggler policies run on the update log (parallel/update_log.py: a replicated log of reduced
gradients; nobody waits for a replica whose gradient is not part of the update):
  Backup(M)     BackupOptimizer: the update fires as soon as M gradients of the current version
                are in (arrival order = one compare-exchange), is the mean of those M, and releases
                everybody; a straggler's late gradient is dropped, the straggler applies the
                update(s) it missed and goes on with fresh parameters
                (reference: backup_worker_optimizer.py:163,241-307). Stragglers still computing
                when the M-th gradient lands short-circuit the rest of their backward pass.
  Interval(ms)  TimeoutReplicasOptimizer.start_interval_updates: a timer on the chief closes a
                window every `interval_ms` and applies the mean of whatever was pushed into it;
                workers push and go on, they never wait (reference:
                DistributedResNet/src/sync_replicas_optimizer_modified/sync_replicas_optimizer_modified.py:207-214,402-430).
                A gradient is tagged with the window that is open when it is pushed, as in the
                reference (local_step is re-read at push time, :293,339); `max_staleness` >= 0 drops
                gradients computed on parameters older than that many updates instead.
  SoftKill(k)   SoftKillOptimizer + short-circuited gradients: once k gradients have arrived the
                remaining replicas stop back-propagating and push zeros for the layers not yet
                done (reference: soft_kill_optimizer.py:424-442, short_circuit_gradients.py:487-545);
                the mean is over all N pushes, zeros included, as in the reference.
  Weighted      least-squares gradient weights under simulated worker kills (reference:
                Distributed_test/src/distributed_train.py:290-341,
                Distributed_test/src/sync_replicas_optimizer_modified/back_up.py:339-349).
The asynchronous policy lives in parallel/async_ps.py.
"""
from __future__ import annotations

import threading
import time
from dataclasses import dataclass
from typing import List, Optional

import numpy as np

from .comm import Communicator
from .update_log import LrLike, lr_at


@dataclass
class StepResult:
    applied: bool            # parameters were updated during this call
    contributors: int        # gradients that entered the (last) update
    aborted: bool            # this replica short-circuited its backward pass
    included: bool           # this replica's gradient was part of an update
    compute_s: float = 0.0   # time from step start to gradients ready on this replica
    global_step: Optional[int] = None   # updates applied so far (None: one more than before iff `applied`)


@dataclass
class PendingGradients:
    """What `compute_gradients` hands to `apply_gradients`: the replica whose arena holds the
    gradient, when the step began, and what the policy already knows about it."""
    replica: object
    t0: float
    done: bool = True          # False: the backward pass was cut short
    compute_s: float = 0.0
    weight: float = 1.0        # Weighted: this replica's least-squares weight for the update
    alive: int = 0             # Weighted: replicas not killed for the update
    total_weight: float = 0.0


class Strategy:
    name = "base"
    needs_abort_points = False
    uses_update_log = False

    def __init__(self, comm: Communicator, board=None):
        self.comm = comm
        self.board = board
        self.world = comm.world
        self.rank = comm.rank
        self.log = None
        self.start_version = 0   # set by the driver after a checkpoint restore, before the first step

    def step(self, replica, step: int, lr: LrLike) -> StepResult:
        """One training step: forward + backward, then aggregation + update."""
        return self.apply_gradients(self.compute_gradients(replica, lr), lr)

    # The two halves of a step, for driver code written like the reference's
    # (grads = opt.compute_gradients(loss); opt.apply_gradients(grads, worker_id, global_step);
    # reference: DistributedLeNet/src/distributed_train.py:188-199).
    def compute_gradients(self, replica, lr: LrLike = 0.0) -> PendingGradients:  # pragma: no cover - interface
        raise NotImplementedError

    def apply_gradients(self, pending: PendingGradients, lr: LrLike) -> StepResult:  # pragma: no cover - interface
        raise NotImplementedError

    def finish(self, collective: bool = True) -> None:
        """End of training (all replicas call it): keeps published data reachable until every
        live replica has caught up, then releases it. collective=False: a worker died, so nothing
        here may wait for all ranks; the survivors agree through the board alone."""
        if self.log is not None:
            self.log.finish(collective=collective) if not collective else self.log.finish()
            if collective:
                self.comm.barrier()
                self.log.close()
            self.log = None

    # helpers --------------------------------------------------------------------------------------
    def _get_log(self, replica):
        if self.log is None:
            from .update_log import make_update_log

            self.log = make_update_log(replica, self.comm, self.board, self.start_version)
        return self.log

    def _plain(self) -> bool:
        """One replica, or no control board: nothing to drop, the step is the plain synchronous one."""
        return self.world == 1 or self.board is None

    def _plain_compute(self, replica) -> PendingGradients:
        """Forward + backward with the bucketed all-reduce overlapped."""
        t0 = time.time()
        with replica.span("forward+backward (gradient all-reduce overlapped)" if self.world > 1 else "forward+backward"):
            replica.forward_backward(on_bucket=self._bucket_reduce(replica))
        return PendingGradients(replica, t0, True, time.time() - t0)

    def _plain_apply(self, pending: PendingGradients, lr: LrLike) -> StepResult:
        replica = pending.replica
        if self.world > 1:
            with replica.span("gradient all-reduce (tail)"), replica.stream_ctx():
                self.comm.wait_all()
        with replica.span("optimizer"):
            replica.apply_update(lr_at(lr, replica.updates_applied), grad_scale=1.0 / self.world)
        return StepResult(True, self.world, False, True, pending.compute_s, replica.updates_applied)

    def _bucket_reduce(self, replica):
        """Callback for forward_backward(on_bucket=...): buckets become final from the end of the
        arena towards offset 0; all but the last overlap the rest of backward on the collective
        stream, the last one (nothing left to overlap with) is issued on the compute stream."""
        grads = replica.arena.grads

        def on_bucket(lo: int, hi: int) -> None:
            if lo == 0:
                self.comm.all_reduce_inline(grads[lo:hi])
            else:
                self.comm.all_reduce_async(grads[lo:hi])
        return on_bucket


class FullSync(Strategy):
    name = "sync"

    def __init__(self, comm: Communicator, board=None, reduce_scatter: bool = False):
        super().__init__(comm, board)
        self.reduce_scatter = reduce_scatter

    def compute_gradients(self, replica, lr: LrLike = 0.0) -> PendingGradients:
        if self.reduce_scatter and self.world > 1:
            t0 = time.time()
            with replica.span("forward+backward"):
                replica.forward_backward()
            return PendingGradients(replica, t0, True, time.time() - t0)
        return self._plain_compute(replica)

    def apply_gradients(self, pending: PendingGradients, lr: LrLike) -> StepResult:
        if self.reduce_scatter and self.world > 1:
            return self._apply_sharded(pending, lr_at(lr, pending.replica.updates_applied))
        return self._plain_apply(pending, lr)

    def _apply_sharded(self, pending: PendingGradients, lr: float) -> StepResult:
        """Reduce-scatter + sharded optimizer + all-gather, ONE collective each way: every rank
        owns an equal contiguous 1/N of the (padded) parameter and optimizer-slot buffers, reduces
        only that range of the gradient, updates it, and the updated ranges are gathered back.
        Moves the same bytes as the all-reduce but the optimizer reads 1/N of the state per rank;
        it is also the layout of the asynchronous parameter shards (parallel/async_ps.py)."""
        replica, dt = pending.replica, pending.compute_s
        a = replica.arena
        shard = a.shard_size(self.world)
        lo, hi = self.rank * shard, (self.rank + 1) * shard
        with replica.span("gradient reduce-scatter"), replica.stream_ctx():
            self.comm.reduce_scatter_flat(a.padded("grads", self.world), shard)
        with replica.span("optimizer (own shard)"):
            replica.apply_update_shard(lr, 1.0 / self.world, lo, hi)
        with replica.span("parameter all-gather"), replica.stream_ctx():
            self.comm.all_gather_flat(a.padded("params", self.world), shard)
        return StepResult(True, self.world, False, True, dt, replica.updates_applied)

    def gather_optimizer_state(self, replica) -> None:
        """Sharded mode keeps each range of the optimizer slots current only on its owner; before a
        checkpoint every replica joins one all-gather per slot buffer so the chief holds all of it."""
        if not (self.reduce_scatter and self.world > 1):
            return
        a = replica.arena
        shard = a.shard_size(self.world)
        with replica.stream_ctx():
            for k in a.slots:
                self.comm.all_gather_flat(a.padded("slot:" + k, self.world), shard)


class _LogStrategy(Strategy):
    uses_update_log = True
    needs_abort_points = True

    def __init__(self, comm: Communicator, board):
        super().__init__(comm, board)
        if board is None and comm.world > 1:
            raise ValueError(f"{self.name} aggregation needs the shared-memory control board")

    def _result(self, res, done: bool, contributors: int, dt: float) -> StepResult:
        return StepResult(res.applied > 0, contributors, not done, res.accepted, dt, res.version)


class Backup(_LogStrategy):
    name = "backup"

    def __init__(self, comm: Communicator, board, replicas_to_aggregate: int):
        super().__init__(comm, board)
        self.m = min(max(1, replicas_to_aggregate), comm.world)

    def compute_gradients(self, replica, lr: LrLike = 0.0) -> PendingGradients:
        if self._plain():
            return self._plain_compute(replica)
        t0 = time.time()
        with replica.span("forward+backward (short-circuits when the update completes)"):
            done = self._get_log(replica).forward_backward(replica)
        return PendingGradients(replica, t0, done, time.time() - t0)

    def apply_gradients(self, pending: PendingGradients, lr: LrLike) -> StepResult:
        if self._plain():
            return self._plain_apply(pending, lr)
        log = self._get_log(pending.replica)
        with pending.replica.span("update log: arrive, reduce, gather + optimizer"):
            if pending.done:
                res = log.contribute(lr, quorum=self.m, abort_at=self.m)
            else:
                res = log.skip(lr)   # the update closed while this replica was still back-propagating
        return self._result(res, pending.done, self.m, pending.compute_s)


class Interval(_LogStrategy):
    name = "interval"
    needs_abort_points = False

    def __init__(self, comm: Communicator, board, interval_ms: float, max_staleness: int = -1):
        super().__init__(comm, board)
        self.interval_s = float(interval_ms) / 1e3
        self.max_staleness = max_staleness
        self._timer: Optional[threading.Thread] = None
        self._timer_stop = threading.Event()
        self._timer_error: Optional[BaseException] = None   # what ended the timer thread
        self.windows_closed = 0
        self._lr: LrLike = 0.0

    def start_interval_updates(self, replica) -> None:
        """Chief only: starts the timer that closes a window every `interval_s` whatever the
        workers are doing (reference: start_interval_updates, sync_replicas_optimizer_modified.py:207-214)."""
        if self._timer is not None or not self.comm.is_chief:
            return
        log = self._get_log(replica)
        import os

        fault_after = int(os.environ.get("DTF_FAULT_INTERVAL_TIMER", "0"))   # tests: the timer fails after this many windows

        def tick() -> None:
            nxt = time.time() + self.interval_s
            while not self._timer_stop.wait(max(0.0, nxt - time.time())):
                nxt += self.interval_s
                try:
                    if fault_after and self.windows_closed >= fault_after:
                        raise RuntimeError("injected fault (DTF_FAULT_INTERVAL_TIMER)")
                    if log.close_window() >= 0:
                        self.windows_closed += 1
                except Exception as e:   # noqa: BLE001 - kept for the training thread (check_timer)
                    self._timer_error = e
                    return

        self._timer = threading.Thread(target=tick, name="interval-updates", daemon=True)
        self._timer.start()

    def check_timer(self) -> None:
        """Every step of the chief's training thread. Without the timer no window ever closes again
        while the workers - which never wait in this mode - go on pushing: the job is stopped and
        the error raised by name instead of a silent stall of the global step."""
        err = self._timer_error
        if err is None and self._timer is not None and not self._timer.is_alive() and not self._timer_stop.is_set():
            err = RuntimeError("thread ended without an error")
        if err is not None:
            self._timer_error = None
            if self.board is not None:
                self.board.report_failure(self.rank)
            raise RuntimeError(f"interval updates (worker {self.rank}): the timer thread failed: {err!r}") from err

    def stop_interval_updates(self) -> None:
        if self._timer is not None:
            self._timer_stop.set()
            self._timer.join(timeout=30)
            self._timer = None

    def _plain(self) -> bool:
        return self.board is None      # one replica with a board still runs on its timer

    def compute_gradients(self, replica, lr: LrLike = 0.0) -> PendingGradients:
        if self._plain():
            return self._plain_compute(replica)
        t0 = time.time()
        log = self._get_log(replica)
        self.start_interval_updates(replica)
        self.check_timer()
        with replica.span("update log: apply closed windows"):
            log.catch_up(lr, reduce_own=False)  # start from the newest parameters, like a worker reading the parameter server
        with replica.span("forward+backward"):
            done = log.forward_backward(replica, abortable=False)
        return PendingGradients(replica, t0, done, time.time() - t0)

    def apply_gradients(self, pending: PendingGradients, lr: LrLike) -> StepResult:
        if self._plain():
            return self._plain_apply(pending, lr)
        log = self._get_log(pending.replica)
        self.check_timer()
        with pending.replica.span("update log: push into the open window"):
            res = log.contribute_open(lr, self.max_staleness)
        return self._result(res, pending.done, log.contributors_hint(), pending.compute_s)

    def finish(self, collective: bool = True) -> None:
        if self.log is not None and collective:
            self.comm.barrier()            # nobody pushes any more
            self.stop_interval_updates()   # chief: no window closes after this
            self.comm.barrier()
        else:
            self.stop_interval_updates()
        super().finish(collective)


class SoftKill(_LogStrategy):
    name = "soft_kill"

    def __init__(self, comm: Communicator, board, should_stop_worker_num: int):
        super().__init__(comm, board)
        self.k = should_stop_worker_num

    def compute_gradients(self, replica, lr: LrLike = 0.0) -> PendingGradients:
        if self._plain():
            return self._plain_compute(replica)
        t0 = time.time()
        with replica.span("forward+backward (short-circuits when the update completes)"):
            done = self._get_log(replica).forward_backward(replica)
        return PendingGradients(replica, t0, done, time.time() - t0)

    def apply_gradients(self, pending: PendingGradients, lr: LrLike) -> StepResult:
        if self._plain():
            return self._plain_apply(pending, lr)
        log = self._get_log(pending.replica)
        # a short-circuited replica still pushes what it has (zeros for the layers it did not reach)
        with pending.replica.span("update log: arrive, reduce, gather + optimizer"):
            res = log.contribute(lr, quorum=self.world, abort_at=self.k, scale=1.0 / self.world)
        return self._result(res, pending.done, self.world, pending.compute_s)


def ls_gradient_weights(num_workers: int, num_batches: int, killed: List[int]) -> np.ndarray:
    """Weights w minimising ||A^T w - 1||_2 where A[r, b] = 1 if worker r covers batch b (every
    worker covers two cyclically adjacent batches) and killed workers' rows are zero
    (reference: Distributed_test/src/distributed_train.py:322-341)."""
    a = np.zeros((num_workers, num_batches), dtype=np.float64)
    for r in range(num_workers):
        a[r, r % num_batches] = 1.0
        a[r, (r + 1) % num_batches] = 1.0
    for r in killed:
        a[r, :] = 0.0
    w, *_ = np.linalg.lstsq(a.T, np.ones(num_batches), rcond=None)
    w[list(killed)] = 0.0
    return w


class Weighted(_LogStrategy):
    name = "weighted"
    needs_abort_points = False

    def __init__(self, comm: Communicator, board, num_worker_kill: int, random_seed: int):
        super().__init__(comm, board)
        self.num_kill = max(0, min(num_worker_kill, comm.world - 1))
        self.seed = random_seed

    def weights_for_step(self, step: int) -> np.ndarray:
        rng = np.random.RandomState(self.seed + step)   # identical on every replica
        killed = list(rng.choice(self.world, size=self.num_kill, replace=False)) if self.num_kill else []
        return ls_gradient_weights(self.world, self.world, killed)

    def compute_gradients(self, replica, lr: LrLike = 0.0) -> PendingGradients:
        if self._plain():
            return self._plain_compute(replica)
        t0 = time.time()
        log = self._get_log(replica)
        # the kill draw belongs to the UPDATE (identical on every replica whatever its iteration count)
        w = self.weights_for_step(log.sync_version())
        wr = float(w[self.rank])
        if wr != 0.0:                            # a killed worker does no work for this update
            log.forward_backward(replica, abortable=False)
        return PendingGradients(replica, t0, True, time.time() - t0, weight=wr, alive=int((w != 0).sum()), total_weight=float(w.sum()))

    def apply_gradients(self, pending: PendingGradients, lr: LrLike) -> StepResult:
        if self._plain():
            return self._plain_apply(pending, lr)
        log = self._get_log(pending.replica)
        if pending.weight == 0.0:
            res = log.skip(lr)
        else:
            total = pending.total_weight
            res = log.contribute(lr, weight=pending.weight, quorum=pending.alive, scale=1.0 / total if total > 0 else 0.0)
        return self._result(res, True, pending.alive, pending.compute_s)


def make_strategy(flags, comm: Communicator, board, replicas_to_aggregate: int) -> Strategy:
    """Strategy switch with the reference's precedence (reference:
    DistributedLeNet/src/distributed_train.py:167-185, DistributedResNet/src/distributed_train.py:238-254)."""
    if getattr(flags, "weighted_grads_method", False):
        return Weighted(comm, board, flags.num_worker_kill, flags.random_seed)
    if flags.interval_method:
        return Interval(comm, board, flags.interval_ms, getattr(flags, "interval_max_staleness", -1))
    if flags.soft_kill_method:
        return SoftKill(comm, board, flags.should_stop_worker_num)
    if flags.backup_worker_method and replicas_to_aggregate < comm.world:
        return Backup(comm, board, replicas_to_aggregate)
    return FullSync(comm, board, getattr(flags, "reduce_scatter", False))
