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
import time
from typing import Dict, List, Optional, Tuple

log = logging.getLogger("dtf")

ITERATION_START_TRACKING = 10   # reference: timeout_manager.py:35
ITERATION_END_TRACKING = 500    # reference: timeout_manager.py:36


def board_name_for(train_dir: str, port: int) -> str:
    base = os.path.basename(os.path.normpath(train_dir)) or "train"
    return f"dtf_{base}_{port}_{os.getuid()}"


def open_board(name: str, world: int, rank: int, nonce: int = 0):
    """Rank 0 creates the segment, the others attach. With a nonce (any value all ranks of THIS run
    agree on) an attacher ignores a segment of the same name left behind by an earlier run."""
    from .. import _host

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
