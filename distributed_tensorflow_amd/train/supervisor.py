"""Chief duties around the training loop: restore-on-start, periodic checkpoints, stop flag.

Equivalent of the `tf.train.Supervisor` use in the reference (reference:
DistributedLeNet/src/distributed_train.py:234-256,280,367-373): the chief restores the newest
checkpoint found in `train_dir` (so re-launching with the same directory resumes), saves every
`save_interval_secs` from a background thread, and saves once more at the end. Non-chief replicas
wait for the chief's parameters (here: one broadcast of the flat arena).

A checkpoint must be a consistent cut, so the training thread takes the host snapshot between
two steps (one D2H copy per flat buffer) when the timer says a save is due; serialising and
writing the bundle happens on the background thread and overlaps with training.
"""
from __future__ import annotations

import logging
import os
import queue
import threading
import time
from typing import Optional

from ..checkpoint.saver import Saver, latest_checkpoint

log = logging.getLogger("dtf")


class Supervisor:
    def __init__(self, saver: Saver, train_dir: str, is_chief: bool, save_interval_secs: float):
        self.saver, self.train_dir, self.is_chief = saver, train_dir, is_chief
        self.save_interval_secs = save_interval_secs
        self._next_save = time.time() + save_interval_secs if save_interval_secs > 0 else float("inf")
        self._q: "queue.Queue" = queue.Queue(maxsize=2)
        self._stop = threading.Event()
        self._thread: Optional[threading.Thread] = None
        self.saves_written = 0
        self.last_saved_step = -1
        if is_chief:
            os.makedirs(train_dir, exist_ok=True)
            self._thread = threading.Thread(target=self._writer, name="checkpoint-writer", daemon=True)
            self._thread.start()

    # -- restore -------------------------------------------------------------------------------------
    def restore_latest(self) -> Optional[int]:
        """Every replica reads the same newest checkpoint (shared directory); returns its step."""
        prefix = latest_checkpoint(self.train_dir)
        if prefix is None:
            return None
        step = self.saver.restore(prefix)
        log.info("Restored %s at global_step %d", prefix, step)
        return step

    # -- periodic save -------------------------------------------------------------------------------
    def should_stop(self) -> bool:
        return self._stop.is_set()

    def request_stop(self) -> None:
        self._stop.set()

    def save_due(self, global_step: int) -> bool:
        return self.is_chief and time.time() >= self._next_save and global_step != self.last_saved_step

    def maybe_save(self, global_step: int, updates_applied: int, force: bool = False) -> bool:
        """Called between steps on the chief; snapshots and queues a save when one is due (or when
        `force`d: the replicas have just gathered sharded optimizer state for this save)."""
        if not self.is_chief or global_step == self.last_saved_step:
            return False
        if not force and time.time() < self._next_save:
            return False
        self._next_save = time.time() + self.save_interval_secs
        snap = self.saver.snapshot(global_step, updates_applied)
        try:
            self._q.put_nowait((global_step, snap))
        except queue.Full:
            return False  # the writer is still busy with the previous checkpoint
        self.last_saved_step = global_step
        return True

    def _writer(self) -> None:
        while True:
            item = self._q.get()
            if item is None:
                return
            step, snap = item
            try:
                self.saver.save(self.train_dir, step, snapshot=snap)
                self.saves_written += 1
            except Exception:  # pragma: no cover - disk errors must not kill training
                log.exception("checkpoint write failed at step %d", step)

    def stop(self) -> None:
        """Waits for queued checkpoints to reach the disk and ends the writer thread."""
        self._stop.set()
        if self._thread is not None:
            self._q.put(None)
            self._thread.join(timeout=120)
            self._thread = None

    def final_save(self, global_step: int, updates_applied: int) -> Optional[str]:
        if not self.is_chief:
            return None
        return self.saver.save(self.train_dir, global_step, updates_applied)
