"""The reference's optimizer-wrapper classes by name, for code that was written against them.

In the reference a synchronisation policy is an `Optimizer` subclass that wraps another optimizer
and wires accumulators and token queues into the TensorFlow graph (reference:
DistributedLeNet/src/backup_worker_experiment/backup_worker_optimizer.py:133-178,
DistributedResNet/src/sync_replicas_optimizer_modified/sync_replicas_optimizer_modified.py:139-214,
DistributedLeNet/src/soft_kill_worker_optimizer/soft_kill_optimizer.py:154-236,
Distributed_test/src/distributed_train.py:165-176). Here the policy is a `Strategy` over a
`Communicator`; these classes keep the reference's constructor arguments and map them onto it:

    comm = Communicator.from_env()                   # one process per GPU (torchrun contract)
    replica = make_replica("resnet", batch_size=128)
    opt = BackupOptimizer("adam", replicas_to_aggregate=6, total_num_replicas=8, comm=comm, board=board)
    result = opt.step(replica, global_step, lr)      # forward + backward + aggregation + update

`opt` (the wrapped optimizer) is the replica's own fused optimizer and is accepted only for
signature compatibility: "adam", "momentum", or an object from optim/flat.py.
"""
from __future__ import annotations

from typing import Optional

from ..parallel.strategies import Backup, FullSync, Interval, SoftKill, StepResult, Strategy, Weighted


class _Wrapper:
    strategy: Strategy

    def __init__(self, opt, total_num_replicas: Optional[int], comm):
        if comm is None:
            raise ValueError("pass comm=Communicator (the process group the replicas synchronise over)")
        if total_num_replicas is not None and total_num_replicas != comm.world:
            raise ValueError(f"total_num_replicas={total_num_replicas} but the process group has {comm.world} ranks")
        self._opt = opt
        self._comm = comm
        self._total_num_replicas = comm.world

    def step(self, replica, global_step: int, lr: float) -> StepResult:
        """One training step of this replica under the policy (the batch must already be loaded
        with `replica.load_batch`). Returns who contributed and whether this replica was cut short."""
        return self.strategy.step(replica, global_step, lr)

    @property
    def total_num_replicas(self) -> int:
        return self._total_num_replicas


class SyncReplicasOptimizer(_Wrapper):
    """tf.train.SyncReplicasOptimizer / SyncReplicasOptimizerV2 as the drivers use them: mean of
    all replicas' gradients (reference: DistributedLeNet/src/distributed_train.py:181-185).
    `replicas_to_aggregate` below the replica count selects backup-worker aggregation."""

    def __init__(self, opt, replicas_to_aggregate: Optional[int] = None, total_num_replicas: Optional[int] = None, *,
                 comm=None, board=None, reduce_scatter: bool = False):
        super().__init__(opt, total_num_replicas, comm)
        m = comm.world if replicas_to_aggregate in (None, -1) else replicas_to_aggregate
        self.replicas_to_aggregate = m
        self.strategy = Backup(comm, board, m) if m < comm.world else FullSync(comm, board, reduce_scatter)


SyncReplicasOptimizerV2 = SyncReplicasOptimizer


class BackupOptimizer(_Wrapper):
    """Mean of the first `replicas_to_aggregate` gradients of a step; the rest are dropped and
    every replica still advances (reference: backup_worker_optimizer.py:163,241-307)."""

    def __init__(self, opt, replicas_to_aggregate: int, total_num_replicas: Optional[int] = None, *,
                 comm=None, board=None):
        super().__init__(opt, total_num_replicas, comm)
        self.replicas_to_aggregate = replicas_to_aggregate
        self.strategy = Backup(comm, board, replicas_to_aggregate)


class TimeoutReplicasOptimizer(_Wrapper):
    """Time-bounded aggregation: an update every `interval_ms` from whatever arrived (reference:
    DistributedResNet/src/sync_replicas_optimizer_modified/sync_replicas_optimizer_modified.py:207-214,402-410).
    The reference arms the timer with `start_interval_updates`; here the interval is a
    constructor argument and the same method only changes it."""

    def __init__(self, opt, global_step: int = 0, total_num_replicas: Optional[int] = None, *,
                 comm=None, board=None, interval_ms: float = 1000.0):
        super().__init__(opt, total_num_replicas, comm)
        self.global_step = global_step
        self.strategy = Interval(comm, board, interval_ms)

    def start_interval_updates(self, interval_ms: float) -> None:
        self.strategy.interval_ns = int(interval_ms * 1e6)


class SoftKillOptimizer(_Wrapper):
    """Stragglers stop back-propagating once `should_stop_worker_num` gradients are in and push
    zeros for the layers they did not reach (reference: soft_kill_optimizer.py:213,424-442,
    short_circuit_gradients.py:487-545)."""

    def __init__(self, opt, global_step: int = 0, total_num_replicas: Optional[int] = None, *,
                 comm=None, board=None, should_stop_worker_num: int = 1):
        super().__init__(opt, total_num_replicas, comm)
        self.global_step = global_step
        self.strategy = SoftKill(comm, board, should_stop_worker_num)


class WeightedGradsOptimizer(_Wrapper):
    """Least-squares weighted aggregation under simulated worker kills (reference:
    Distributed_test/src/distributed_train.py:290-341)."""

    def __init__(self, opt, replicas_to_aggregate: Optional[int] = None, total_num_replicas: Optional[int] = None, *,
                 comm=None, board=None, num_worker_kill: int = 3, random_seed: int = 142):
        super().__init__(opt, total_num_replicas, comm)
        self.strategy = Weighted(comm, board, num_worker_kill, random_seed)
