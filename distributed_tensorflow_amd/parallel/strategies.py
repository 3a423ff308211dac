"""Gradient-aggregation policies: what the reference implements as TensorFlow optimizer wrappers
around parameter-server accumulators and token queues, re-expressed over collectives.

Every policy ends a step with ALL replicas holding identical parameters: each replica runs the
same fused optimizer on the same aggregated gradient, so no parameter broadcast is needed.

  FullSync      tf.train.SyncReplicasOptimizer(V2): mean of all N gradients
                (reference: DistributedLeNet/src/distributed_train.py:181-185).
  Backup(M)     BackupOptimizer: mean of the first M gradients of the step, the other N-M are
                dropped, everybody advances (reference: backup_worker_optimizer.py:163,241-307).
                Realised as a MASKED all-reduce: arrival order comes from one atomic fetch-add on
                the shared-memory board; a replica whose order >= M contributes zeros; stragglers
                still computing when the M-th gradient lands short-circuit their backward pass so
                the collective is not held up by them; divisor M.
  Interval(ms)  TimeoutReplicasOptimizer.start_interval_updates: the update uses whatever arrived
                within `interval_ms` of the step opening (reference:
                DistributedResNet/src/sync_replicas_optimizer_modified/sync_replicas_optimizer_modified.py:207-214,402-410);
                late replicas short-circuit and contribute zeros; divisor = number in time
                (carried through the same collective); no gradient in time = no update.
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

import time
from dataclasses import dataclass
from typing import List

import numpy as np

from .comm import Communicator


@dataclass
class StepResult:
    applied: bool            # parameters were updated
    contributors: int        # gradients that entered the update
    aborted: bool            # this replica short-circuited its backward pass
    included: bool           # this replica's gradient was part of the update
    compute_s: float = 0.0   # time from step start to gradients ready on this replica


class Strategy:
    name = "base"
    needs_abort_points = False

    def __init__(self, comm: Communicator, board=None):
        self.comm = comm
        self.board = board
        self.world = comm.world
        self.rank = comm.rank

    def step(self, replica, step: int, lr: float) -> StepResult:  # pragma: no cover - interface
        raise NotImplementedError

    # helpers --------------------------------------------------------------------------------------
    def _allreduce_full(self, replica, with_aux: bool = False) -> None:
        a = replica.arena
        with replica.stream_ctx():   # collectives order against the replica's compute stream
            self.comm.all_reduce_inline(a.grads_full if with_aux else a.grads)
            self.comm.wait_all()

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

    def step(self, replica, step: int, lr: float) -> StepResult:
        t0 = time.time()
        if self.reduce_scatter and self.world > 1:
            return self._step_sharded(replica, lr, t0)
        replica.forward_backward(on_bucket=self._bucket_reduce(replica))
        dt = time.time() - t0
        with replica.stream_ctx():
            self.comm.wait_all()
        replica.apply_update(lr, grad_scale=1.0 / self.world)
        return StepResult(True, self.world, False, True, dt)

    def _step_sharded(self, replica, lr: float, t0: float) -> StepResult:
        """Reduce-scatter + sharded optimizer + all-gather: every rank owns a contiguous 1/N of
        the parameters and of the optimizer slots, reduces only that range of the gradient,
        updates it, and the updated ranges are gathered back. Moves the same bytes as the
        all-reduce but the optimizer reads 1/N of the state per rank; it is also the layout of
        the asynchronous parameter shards (parallel/async_ps.py)."""
        a = replica.arena
        bounds = a.shard_bounds(self.world)
        replica.forward_backward()
        dt = time.time() - t0
        lo, hi = bounds[self.rank]
        with replica.stream_ctx():
            self.comm.reduce_scatter_flat(a.grads, bounds)
        replica.apply_update_shard(lr, 1.0 / self.world, lo, hi)
        with replica.stream_ctx():
            self.comm.all_gather_flat(a.params, bounds)
        return StepResult(True, self.world, False, True, dt)


class Backup(Strategy):
    name = "backup"
    needs_abort_points = True

    def __init__(self, comm: Communicator, board, replicas_to_aggregate: int):
        super().__init__(comm, board)
        if board is None and comm.world > 1:
            raise ValueError("backup-worker aggregation needs the shared-memory control board")
        self.m = min(max(1, replicas_to_aggregate), comm.world)

    def step(self, replica, step: int, lr: float) -> StepResult:
        b, m = self.board, self.m
        t0 = time.time()
        if self.world == 1 or m >= self.world:
            # nothing can be dropped: plain synchronous step
            replica.forward_backward(on_bucket=self._bucket_reduce(replica))
            dt = time.time() - t0
            with replica.stream_ctx():
                self.comm.wait_all()
            replica.apply_update(lr, grad_scale=1.0 / self.world)
            return StepResult(True, self.world, False, True, dt)
        b.open_step(step)
        done = replica.forward_backward(should_abort=lambda: b.stop_requested(step))
        replica.synchronize()
        dt = time.time() - t0
        if done:
            order = b.arrive(step, self.rank, m)
            included = order < m
        else:
            b.arrive_aborted(step, self.rank)
            included = False
        if not included:
            replica.zero_grads()
        self._allreduce_full(replica)
        replica.apply_update(lr, grad_scale=1.0 / m)
        return StepResult(True, m, not done, included, dt)


class Interval(Strategy):
    name = "interval"
    needs_abort_points = True

    def __init__(self, comm: Communicator, board, interval_ms: float):
        super().__init__(comm, board)
        if board is None and comm.world > 1:
            raise ValueError("interval aggregation needs the shared-memory control board")
        self.interval_ns = int(interval_ms * 1e6)

    def step(self, replica, step: int, lr: float) -> StepResult:
        b = self.board
        t0 = time.time()
        if b is not None:
            start_ns = b.open_step(step)
            now = b.now_ns
        else:
            start_ns = time.time_ns()
            now = time.time_ns
        deadline = start_ns + self.interval_ns
        done = replica.forward_backward(should_abort=lambda: now() > deadline)
        replica.synchronize()
        dt = time.time() - t0
        in_time = done and now() <= deadline
        if not in_time:
            replica.zero_grads()
        aux = replica.arena.aux
        with replica.stream_ctx():
            aux.zero_()
            aux[0] = 1.0 if in_time else 0.0
        self._allreduce_full(replica, with_aux=True)
        replica.synchronize()
        count = int(round(float(aux[0].item())))
        if count > 0:
            replica.apply_update(lr, grad_scale=1.0 / count)
        return StepResult(count > 0, count, not done, in_time, dt)


class SoftKill(Strategy):
    name = "soft_kill"
    needs_abort_points = True

    def __init__(self, comm: Communicator, board, should_stop_worker_num: int):
        super().__init__(comm, board)
        if board is None and comm.world > 1:
            raise ValueError("soft-kill aggregation needs the shared-memory control board")
        self.k = should_stop_worker_num

    def step(self, replica, step: int, lr: float) -> StepResult:
        b = self.board
        t0 = time.time()
        if b is not None:
            b.open_step(step)
            done = replica.forward_backward(should_abort=lambda: b.stop_requested(step))
        else:
            done = replica.forward_backward()
        replica.synchronize()
        dt = time.time() - t0
        if b is not None:
            if done:
                b.arrive(step, self.rank, self.k)   # raises the stop flag on the k-th arrival
            else:
                b.arrive_aborted(step, self.rank)
        self._allreduce_full(replica)
        replica.apply_update(lr, grad_scale=1.0 / self.world)
        return StepResult(True, self.world, not done, True, dt)


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


class Weighted(Strategy):
    name = "weighted"

    def __init__(self, comm: Communicator, board, num_worker_kill: int, random_seed: int):
        super().__init__(comm, board)
        self.num_kill = max(0, min(num_worker_kill, comm.world - 1))
        self.seed = random_seed

    def weights_for_step(self, step: int) -> np.ndarray:
        rng = np.random.RandomState(self.seed + step)   # identical on every replica
        killed = list(rng.choice(self.world, size=self.num_kill, replace=False)) if self.num_kill else []
        return ls_gradient_weights(self.world, self.world, killed)

    def step(self, replica, step: int, lr: float) -> StepResult:
        w = self.weights_for_step(step)
        wr = float(w[self.rank])
        t0 = time.time()
        if wr == 0.0:
            # a killed worker does no work this iteration
            replica.zero_grads()
            done = False
        else:
            done = replica.forward_backward()
            replica.scale_grads(wr)
        dt = time.time() - t0
        self._allreduce_full(replica)
        total = float(w.sum())
        replica.apply_update(lr, grad_scale=1.0 / total if total > 0 else 0.0)
        return StepResult(total > 0, int((w != 0).sum()), False, wr != 0.0, dt)


def make_strategy(flags, comm: Communicator, board, replicas_to_aggregate: int) -> Strategy:
    """Strategy switch with the reference's precedence (reference:
    DistributedLeNet/src/distributed_train.py:167-185, DistributedResNet/src/distributed_train.py:238-254)."""
    if getattr(flags, "weighted_grads_method", False):
        return Weighted(comm, board, flags.num_worker_kill, flags.random_seed)
    if flags.interval_method:
        return Interval(comm, board, flags.interval_ms)
    if flags.soft_kill_method:
        return SoftKill(comm, board, flags.should_stop_worker_num)
    if flags.backup_worker_method and replicas_to_aggregate < comm.world:
        return Backup(comm, board, replicas_to_aggregate)
    return FullSync(comm, board, getattr(flags, "reduce_scatter", False))
