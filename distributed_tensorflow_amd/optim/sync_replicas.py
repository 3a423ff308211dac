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

or, in the two phases of the reference's drivers (DistributedLeNet/src/distributed_train.py:188-207):

    grads = opt.compute_gradients(replica)           # forward + backward (short-circuiting where the policy has it)
    result = opt.apply_gradients(grads, worker_id, global_step, lr=lr)   # push / aggregate / update
    opt.get_chief_queue_runner().start(); opt.get_init_tokens_op()()     # chief-side plumbing, same call sites

`opt` (the wrapped optimizer) is the replica's own fused optimizer and is accepted only for
signature compatibility: "adam", "momentum", or an object from optim/flat.py.
"""
from __future__ import annotations

from typing import Callable, Optional

from ..parallel.strategies import Backup, FullSync, Interval, PendingGradients, SoftKill, StepResult, Strategy, Weighted


class ChiefQueueRunner:
    """What the reference starts on the chief with `sv.start_queue_runners(sess, [chief_queue_runner])`:
    the thread that applies updates (reference: backup_worker_optimizer.py:314-315,
    DistributedLeNet/src/distributed_train.py:260-262). Updates are applied by the replicas themselves
    here, so only the timer-driven policy has a chief thread to start; for the others `start` has
    nothing to do and `threads` stays empty."""

    def __init__(self, start: Optional[Callable[[], None]] = None, stop: Optional[Callable[[], None]] = None):
        self._start, self._stop = start, stop
        self.started = False

    def start(self) -> "ChiefQueueRunner":
        if not self.started and self._start is not None:
            self._start()
        self.started = True
        return self

    def stop(self) -> None:
        if self.started and self._stop is not None:
            self._stop()
        self.started = False


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

    # ---- the reference's two-phase Optimizer API --------------------------------------------------
    def compute_gradients(self, replica, lr: float = 0.0) -> PendingGradients:
        """Forward + backward of `replica` on the batch it was given; the gradient stays in the
        replica's flat arena (the returned handle stands for the reference's grads_and_vars list).
        Policies that short-circuit back-propagation do it here (reference:
        soft_kill_optimizer.py:249-273). `lr` is only used by the timer-driven policy, which first
        applies the updates closed since the last step."""
        self._replica = replica
        return self.strategy.compute_gradients(replica, lr)

    def apply_gradients(self, grads: PendingGradients, worker_id: Optional[int] = None, global_step: Optional[int] = None, *,
                        lr: float) -> StepResult:
        """Pushes the gradient into the policy's aggregation and applies whatever update(s) result
        (reference: apply_gradients(grads, worker_id, global_step), backup_worker_optimizer.py:191-313).
        `worker_id` must be this process's rank; `global_step` is informational - the policy counts
        the updates itself and returns the new value in the result."""
        if worker_id is not None and worker_id != self._comm.rank:
            raise ValueError(f"worker_id={worker_id} but this process is rank {self._comm.rank}")
        return self.strategy.apply_gradients(grads, lr)

    def get_chief_queue_runner(self) -> ChiefQueueRunner:
        st = self.strategy
        if isinstance(st, Interval):
            return ChiefQueueRunner(lambda: st.start_interval_updates(self._replica), st.stop_interval_updates)
        return ChiefQueueRunner()

    def get_init_tokens_op(self, num_tokens: int = -1) -> Callable[[], None]:
        """Callable that seeds one token per replica (reference: get_init_tokens_op; the drivers run
        it once on the chief). Tokens only gate the worker-time-CDF protocol here (`_wait_op`)."""
        board, world = self.strategy.board, self._comm.world
        n = 1 if num_tokens < 0 else max(1, num_tokens // world)

        def op() -> None:
            if board is not None:
                for r in range(world):
                    board.put_tokens(r, n, board.global_step())
        return op

    def _wait_op(self, timeout_s: float = 600.0):
        """Takes this replica's token (blocks until the chief put one); returns the step it carries."""
        board = self.strategy.board
        return None if board is None else board.take_token(self._comm.rank, timeout_s)

    # Initialisation ops of the reference's graph; replicas here start from broadcast parameters and
    # the policies keep their own step counters, so these exist for call-site compatibility.
    @property
    def chief_init_op(self) -> Callable[[], None]:
        return lambda: None

    @property
    def local_step_init_op(self) -> Callable[[], None]:
        return lambda: None

    @property
    def ready_for_local_init_op(self) -> Callable[[], bool]:
        return lambda: True

    def finish(self) -> None:
        self.strategy.finish()


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

    def start_interval_updates(self, interval_ms: Optional[float] = None, replica=None) -> None:
        """Chief: arms the timer (reference: start_interval_updates(sess, timeout_client),
        sync_replicas_optimizer_modified.py:207-214). Non-chief replicas: no-op."""
        if interval_ms is not None:
            self.strategy.interval_s = float(interval_ms) / 1e3
        replica = replica if replica is not None else getattr(self, "_replica", None)
        if replica is not None:
            self.strategy.start_interval_updates(replica)


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

    def make_session_run_hook(self, is_chief: bool, num_tokens: int = -1):
        """Reference: make_session_run_hook -> _SyncReplicasOptimizerHook (begin /
        after_create_session: local init, chief seeds the tokens and starts the queue runner;
        weighted_grad_apply_syncReplicOpt.py:398-451). Returns an object with the same two methods."""
        opt = self

        class _Hook:
            def begin(self) -> None:
                self.runner = opt.get_chief_queue_runner() if is_chief else None

            def after_create_session(self, session=None, coord=None) -> None:
                opt.local_step_init_op()
                if is_chief:
                    opt.chief_init_op()
                    opt.get_init_tokens_op(num_tokens)()
                    self.runner.start()
        return _Hook()
