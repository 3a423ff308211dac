"""Control plane: startup barrier, per-worker timing telemetry, update notifications.

The reference runs a Twisted Perspective-Broker server in every worker and a client that
broadcasts to all of them (reference: DistributedLeNet/src/timeout_manager.py): a ready handshake
that blocks training until every worker is up (`launch_manager`, :198-211),
`worker_dequeued_token` / `worker_finished_computing_gradients` timestamps from which worker 0
logs the sorted `ELAPSED TIMES` and the `ITERATION TIMES` every 50 iterations for iterations
10..500 (:48-70), and `parameters_updated` notifications (:38-46).

On one node all of that is shared memory: the `Board` (csrc/host/board.cc) holds ready flags and
a ring of per-(worker, iteration) timestamps, so a "broadcast" is one store and the chief reads
everybody's times directly. The log lines keep the reference's exact text because its benchmark
tooling parses them (reference: DistributedLeNet/tools/benchmark.py:60-144).
"""
from __future__ import annotations

import logging
import os
import sys
import threading
import time
from typing import Callable, Dict, List, Optional, Set, Tuple

log = logging.getLogger("dtf")

ITERATION_START_TRACKING = 10   # reference: timeout_manager.py:35
ITERATION_END_TRACKING = 500    # reference: timeout_manager.py:36


def board_name_for(train_dir: str, port: int) -> str:
    base = os.path.basename(os.path.normpath(train_dir)) or "train"
    return f"dtf_{base}_{port}_{os.getuid()}"


def open_board(name: str, world: int, rank: int, nonce: int = 0, kind: Optional[str] = None):
    """Rank 0 creates the segment, the others attach. With a nonce (any value all ranks of THIS run
    agree on) an attacher ignores a segment of the same name left behind by an earlier run.
    kind "store" (or DTF_CONTROL_PLANE=store): the same interface over the process group's key-value
    store, for workers on several machines (parallel/store_board.py)."""
    from .. import _host

    kind = kind or os.environ.get("DTF_CONTROL_PLANE", "shm")
    if kind == "store":
        from .store_board import StoreBoard, default_store

        return StoreBoard(default_store(), "%s_%x" % (name, nonce), world, rank)
    if kind != "shm":
        raise ValueError("control plane must be shm or store")
    return _host.Board(name, world, rank == 0, nonce)


def run_nonce(comm) -> int:
    """A value unique to this run, agreed by all ranks through the process group."""
    import random

    mine = random.SystemRandom().getrandbits(62) | 1
    return int(comm.all_gather_object(mine)[0])


def unlink_on_sigterm(name: str) -> None:
    """The launcher stops workers with SIGTERM, which skips destructors: remove the segment first,
    then let the default action end the process."""
    import signal

    from .. import _host

    def handler(signum, frame):
        try:
            _host.Board.unlink(name)
        finally:
            signal.signal(signum, signal.SIG_DFL)
            os.kill(os.getpid(), signum)

    try:
        signal.signal(signal.SIGTERM, handler)
    except ValueError:
        pass   # not the main thread


class TimeoutClient:
    """Broadcast side (same method names as the reference's client)."""

    def __init__(self, board, worker_id: int, n_workers: int):
        self.board, self.worker_id, self.n = board, worker_id, n_workers
        self._dequeue_ns: Dict[int, int] = {}

    def broadcast_worker_dequeued_token(self, iteration: int) -> None:
        self._dequeue_ns[iteration] = self.board.now_ns() if self.board is not None else time.time_ns()
        log.info("Worker %d dequeued token on iteration %d - %d", self.worker_id, iteration, time.time())

    def broadcast_worker_finished_computing_gradients(self, iteration: int) -> None:
        now = self.board.now_ns() if self.board is not None else time.time_ns()
        t0 = self._dequeue_ns.pop(iteration, now)
        if self.board is not None:
            self.board.post_times(self.worker_id, iteration, t0, now)
        log.info("Worker %d finished computing gradients on iteration %d - %d", self.worker_id, iteration, time.time())

    def broadcast_parameters_updated(self, step: int) -> None:
        log.info("Parameters have been updated on step %d..", step)

    def ready_to_start(self) -> bool:
        return self.board is None or self.board.all_ready()


class TimeoutServer:
    """Collecting side; on worker 0 it produces the ELAPSED / ITERATION TIMES report."""

    def __init__(self, board, worker_id: int, n_workers: int):
        self.board, self.worker_id, self.n = board, worker_id, n_workers
        self.compute_times: List[Tuple[int, float, int]] = []     # (worker, elapsed, iteration)
        self.iteration_start_times: Dict[int, float] = {}
        self._collected = set()

    def collect(self, iteration: int) -> bool:
        """Pulls every worker's timestamps of `iteration`; True once all are in."""
        if self.board is None or iteration in self._collected:
            return True
        recs = []
        for w in range(self.n):
            r = self.board.get_times(w, iteration)
            if r is None:
                return False
            recs.append((w, r))
        for w, (t0, t1) in recs:
            self.compute_times.append((w, (t1 - t0) / 1e9, iteration))
        self.iteration_start_times[iteration] = min(t0 for _, (t0, _) in recs) / 1e9
        self._collected.add(iteration)
        return True

    def maybe_report(self, iteration: int) -> Optional[Tuple[list, list]]:
        """Logs the two reference lines on the reference's cadence; returns what was logged."""
        if self.worker_id != 0 or iteration <= ITERATION_START_TRACKING:
            return None
        if not (iteration % 50 == 0 or iteration == ITERATION_END_TRACKING):
            return None
        elapsed_times = sorted([(x[1], x[0], x[2]) for x in self.compute_times if x[2] > ITERATION_START_TRACKING], key=lambda x: x[0])
        starts = [t for i, t in sorted(self.iteration_start_times.items()) if i > ITERATION_START_TRACKING]
        iteration_times = [starts[i + 1] - starts[i] for i in range(len(starts) - 1)]
        log.info("ELAPSED TIMES %s" % str(elapsed_times))
        log.info("ITERATION TIMES %s" % str(iteration_times))
        return elapsed_times, iteration_times


def launch_manager(board, worker_id: int, n_workers: int, timeout_s: float = 600.0):
    """Marks this worker ready and blocks until all workers are (the reference's startup
    barrier). Returns (client, server)."""
    client = TimeoutClient(board, worker_id, n_workers)
    server = TimeoutServer(board, worker_id, n_workers)
    log.info("Worker %d: starting status server...", worker_id)
    if board is not None:
        board.mark_ready(worker_id)
        if not board.wait_all_ready(timeout_s):
            raise TimeoutError(f"worker {worker_id}: not all {n_workers} workers became ready within {timeout_s}s")
    log.info("Server ready to start!")
    return client, server


class Watchdog(threading.Thread):
    """Liveness of the workers of one job.

    The reference's side channel keeps a reconnecting client per worker with a 30 s protocol
    timeout (reference: DistributedLeNet/src/timeout_manager.py:79-105,120-130): a worker that
    goes away is noticed by the others. Here every worker runs this thread: it stamps the worker's
    own heartbeat on the board (so the stamp says "the process is alive", whatever its training
    thread is blocked in) and reads everybody else's. A worker whose stamp is older than
    `silence_s` is declared dead, by name, once:

      * `tolerate(dead)` true - the policy can go on without it (backup workers with enough
        replicas left, timer-driven updates while the chief lives): the worker is marked dead on
        the board, which takes it out of the update log's back-pressure, and `on_dead(rank)` runs;
      * otherwise the job cannot make progress (a collective would wait for the dead worker until
        the process group's own timeout): the stop flag is raised for everybody and this process
        ends at once with `exit_code` - from the thread, because the training thread may be stuck
        inside a collective. Nothing is re-executed in a process that holds a GPU; the launcher
        sees the non-zero status.

    A worker that finishes normally calls `depart()`, which marks it as gone on purpose.
    """

    def __init__(self, board, rank: int, world: int, silence_s: float, tolerate: Optional[Callable[[Set[int]], bool]] = None,
                 on_dead: Optional[Callable[[int], None]] = None, exit_code: int = 3, period_s: Optional[float] = None):
        super().__init__(name="liveness-watchdog", daemon=True)
        self.board, self.rank, self.world = board, rank, world
        self.silence_s = float(silence_s)
        self.period_s = period_s if period_s is not None else max(0.05, min(1.0, self.silence_s / 8))
        self.tolerate = tolerate or (lambda dead: False)
        self.on_dead = on_dead
        self.exit_code = exit_code
        self.dead: Set[int] = set()
        self._stop_evt = threading.Event()
        self._t_start_ns = 0

    def run(self) -> None:
        b = self.board
        self._t_start_ns = b.now_ns()
        while not self._stop_evt.wait(self.period_s):
            b.heartbeat(self.rank)
            if b.should_stop():
                failed = b.failed_rank()
                if failed >= 0 and failed != self.rank:
                    # that worker's training thread raises the error itself; this one may be waiting
                    # for it in a collective or a bounded device wait, so the process ends from here
                    log.error("Worker %d: worker %d reported a failed helper thread and stopped the job (see its log); stopping",
                              self.rank, failed)
                    self._exit_now()
                # a normal end of the run - or another worker's watchdog was faster: its verdict is on the board
                mask = b.agg_dead()
                gone = {r for r in range(self.world) if (mask >> r) & 1}
                if gone - self.dead and not self.tolerate(gone | self.dead):
                    for r in sorted(gone - self.dead):
                        log.error("Worker %d: worker %d has been silent and is taken for dead (declared by another worker); "
                                  "the job cannot continue, stopping", self.rank, r)
                    self._exit_now()
                return
            now = b.now_ns()
            for r in range(self.world):
                if r == self.rank or r in self.dead:
                    continue
                last = b.last_heartbeat_ns(r)
                if last == type(b).DEPARTED:
                    continue
                silent_s = (now - max(last, self._t_start_ns)) / 1e9
                if silent_s > self.silence_s:
                    self._declare_dead(r, silent_s)

    def _declare_dead(self, r: int, silent_s: float) -> None:
        self.dead.add(r)
        b = self.board
        b.agg_mark_dead(r)
        if self.tolerate(set(self.dead)):
            log.error("Worker %d: worker %d has been silent for %.1f s and is taken for dead; continuing without it",
                      self.rank, r, silent_s)
            if self.on_dead is not None:
                try:
                    self.on_dead(r)
                except Exception as e:   # noqa: BLE001 - the watchdog must keep running
                    log.error("Worker %d: on_dead(%d) failed: %s", self.rank, r, e)
            return
        log.error("Worker %d: worker %d has been silent for %.1f s and is taken for dead; the job cannot continue, stopping",
                  self.rank, r, silent_s)
        b.request_stop()
        self._exit_now()

    def _exit_now(self) -> None:
        for h in log.handlers + logging.getLogger().handlers:
            try:
                h.flush()
            except Exception:   # noqa: BLE001
                pass
        sys.stderr.flush()
        os._exit(self.exit_code)

    def beat(self) -> None:
        self.board.heartbeat(self.rank)

    def depart(self) -> None:
        """This worker is done: stop watching and tell the others it left on purpose."""
        self._stop_evt.set()
        self.board.depart(self.rank)
