"""Update log over peer-mapped HBM (csrc/comm/update_log.hip): the device implementation of the
protocol in `parallel/update_log.py`.

One GPU per process, one node. Every replica owns a fine-grained control block, a staging buffer
and a ring of reduced vectors, exports them as HIP IPC handles and maps everybody else's. A step is
two stream-ordered launches (`ulog_begin`, one thread; `ulog_step`, a co-resident grid): arrival,
quorum, reduction of the own shard, gathering and the fused optimizer all happen on the device, so
the host never synchronises with the GPU on the training path and never learns - or needs to learn -
whether its gradient made it into the update (reference semantics: the first M fresh gradients make
the update and everybody is released, DistributedLeNet/src/backup_worker_experiment/
backup_worker_optimizer.py:241-307; timer-driven windows, DistributedResNet/src/
sync_replicas_optimizer_modified/sync_replicas_optimizer_modified.py:207-214).

The kernels mirror eight counters (version, accepted, dropped, ...) into pinned host memory after
every launch; `ContributionResult`s are built from that mirror. `sync_status=True` waits for the
launch before reading it (a driver that reads the loss back every step pays nothing extra);
otherwise the numbers are those of the newest finished launch and `stale` says so.
"""
from __future__ import annotations

import os
import threading
import time
from typing import List, Optional

import torch

from ..lr import schedule_params
from ..ops import native as nv
from .update_log import ContributionResult, LrLike

MODE_CONTRIBUTE, MODE_OPEN, MODE_CATCH_UP, MODE_SKIP = 0, 1, 2, 3
US_SEQ, US_VERSION, US_ACCEPTED, US_DROPPED, US_APPLIED, US_LAST, US_ERROR, US_CONTRIB = range(8)

_ERRORS = {
    1: "the slowest live replica did not catch up in time (back-pressure)",
    2: "the blocks of an aggregation launch did not all start (not co-resident)",
    3: "this replica fell more than the ring of kept updates behind",
    4: "the update this replica contributed to never closed",
    5: "an owner never published its reduced shard",
    6: "a window stayed frozen",
}


def alloc_control_block() -> int:
    """A zeroed fine-grained control block (device pointer). A replica that wants its kernels to
    short-circuit allocates it BEFORE recording its step program and hands `abort_words(block)` to
    the recorder; the update log created later adopts the same block."""
    h = nv.hip()
    return int(h.ipc_alloc(int(h.ULOG_CTL_BYTES), True))


def abort_words(block: int):
    """(stop word, version word) addresses inside a control block, for `_hip.set_abort_ref`."""
    h = nv.hip()
    wb = int(h.ULOG_WORD_BYTES)
    return block + int(h.UW_ABORT) * wb, block + int(h.UW_VERSION) * wb


class DeviceLogUnavailable(RuntimeError):
    """Raised on EVERY rank when some rank could not map a peer's buffers."""


class DeviceUpdateLog:
    """Update log for `HipReplica`s of one node."""

    TICKS_PER_SECOND = 100_000_000   # wall_clock64() runs at 100 MHz

    def __init__(self, replica, comm, board, name: str, timeout_s: float = 600.0, start_version: int = 0,
                 blocks: int = 32, sync_status: bool = True):
        nv.require()
        h = nv.hip()
        if comm.world > 8:
            raise ValueError("the device update log covers one node (at most 8 GPUs)")
        self.replica, self.comm, self.board = replica, comm, board
        self.rank, self.world = comm.rank, comm.world
        self.ring = int(h.ULOG_RING)
        self.total = replica.arena.total
        self.blocks = blocks
        self.sync_status = sync_status
        # an in-kernel wait is bounded well below the watchdogs that look at the GPU from outside
        self.timeout_ticks = int(min(timeout_s, 20.0) * self.TICKS_PER_SECOND)
        self.max_version = 1 << 60
        self._lock = threading.Lock()
        self._launched = 0
        self._last_lr: LrLike = 0.0
        self._closed = False
        self.stream = replica.stream
        self.close_stream: Optional[torch.cuda.Stream] = None
        self._ctl_stream: Optional[torch.cuda.Stream] = None
        dev = replica.device
        # control block: the replica's own if its step program short-circuits on it
        self._own_ctl = getattr(replica, "ctl_block", 0) or alloc_control_block()
        self._ctl_is_replicas = bool(getattr(replica, "ctl_block", 0))
        self._own_stage = int(h.ipc_alloc(self.total * 4, True))
        self._own_red = int(h.ipc_alloc(self.total * 4 * self.ring, True))
        h.ulog_init_ctl(self._own_ctl, int(start_version), self.stream.cuda_stream)
        self.stream.synchronize()
        handles = comm.all_gather_object(tuple(h.ipc_export(p) for p in (self._own_ctl, self._own_stage, self._own_red)))
        self._ctl: List[int] = []
        self._stage: List[int] = []
        self._red: List[int] = []
        self._imported: List[int] = []
        # Mapping a peer's HBM can fail (ranks on different nodes, a GPU pair without peer access):
        # the ranks agree on the outcome, so that either all of them use this log or none does.
        error = None
        try:
            if os.environ.get("DTF_ULOG_FAIL_IMPORT") == str(self.rank):   # test hook: this rank cannot map its peers
                raise RuntimeError("peer mapping disabled by DTF_ULOG_FAIL_IMPORT")
            for r, hs in enumerate(handles):
                if r == self.rank:
                    ptrs = (self._own_ctl, self._own_stage, self._own_red)
                else:
                    ptrs = []
                    for x in hs:
                        ptrs.append(int(h.ipc_import(x)))
                        self._imported.append(ptrs[-1])
                self._ctl.append(ptrs[0])
                self._stage.append(ptrs[1])
                self._red.append(ptrs[2])
        except Exception as e:   # reported below, on every rank
            error = f"rank {self.rank}: {e!r}"
        errors = [e for e in comm.all_gather_object(error) if e]
        if errors:
            self.close()
            raise DeviceLogUnavailable("; ".join(errors))
        self.status = torch.zeros(8, dtype=torch.int64).pin_memory()
        self.status[US_VERSION] = int(start_version)
        self.close_status = torch.zeros(8, dtype=torch.int64).pin_memory()
        self._status_np = self.status.numpy()
        self._close_np = self.close_status.numpy()
        self._device = dev
        comm.barrier()

    # ---------------------------------------------------------------------------------------------
    @property
    def version(self) -> int:
        """Updates this replica has applied, as of the newest finished launch."""
        return int(self._status_np[US_VERSION])

    @property
    def dropped(self) -> int:
        return int(self._status_np[US_DROPPED])

    @property
    def contributed(self) -> int:
        return int(self._status_np[US_ACCEPTED])

    def wait_status(self) -> None:
        """Waits for every launch issued so far and raises if a wait inside one timed out."""
        self.stream.synchronize()
        self.check()

    def check(self) -> None:
        err = int(self._status_np[US_ERROR])
        if err:
            raise TimeoutError(f"update log (rank {self.rank}, version {self.version}): {_ERRORS.get(err, f'error {err}')}")

    def sync_version(self) -> int:
        self.wait_status()
        return self.version

    def global_version(self) -> int:
        """The update that is open now (reads the coordinator's word; synchronous, not for the training path)."""
        w = nv.hip().ulog_peek(self._ctl[0], int(nv.hip().UW_OPEN), 1)[0]
        return int(w >> 32)

    def open_count(self) -> int:
        w = nv.hip().ulog_peek(self._ctl[0], int(nv.hip().UW_OPEN), 1)[0]
        return int((w >> 16) & 0x7FFF)

    def entry(self, u: int):
        """(count, mask, owners, scale) of closed update u as seen in this replica's block, or None."""
        h = nv.hip()
        slot = u % self.ring
        tag = h.ulog_peek(self._own_ctl, int(h.UW_HIST_TAG) + slot, 1)[0]
        aux = h.ulog_peek(self._own_ctl, int(h.UW_HIST_AUX) + slot, 1)[0]
        if (tag >> 32) - 1 != u:
            return None
        import struct

        scale = struct.unpack("<f", struct.pack("<I", aux & 0xFFFFFFFF))[0]
        return int((tag >> 16) & 0xFFFF), int(tag & 0xFFFF), int((aux >> 32) & 0xFFFF), scale

    # ---------------------------------------------------------------------------------------------
    def _args(self, lr: LrLike, mode: int, *, weight: float = 1.0, quorum: int = 1, abort_at: int = 0, scale: float = 0.0,
              max_staleness: int = -1, status: Optional[torch.Tensor] = None) -> dict:
        r, a = self.replica, self.replica.arena
        lr0, steps, decay = schedule_params(lr)
        adam = r.optimizer == "adam"
        return dict(
            ctl=self._ctl, stage=self._stage, red=self._red, grads=a.grads.data_ptr(), p=a.params.data_ptr(),
            m=a.slots["m"].data_ptr(), v=a.slots["v"].data_ptr() if adam else None,
            status=(self.status if status is None else status).data_ptr(), n=self.total, rank=self.rank, world=self.world,
            mode=mode, quorum=int(quorum), abort_at=int(abort_at), scale=float(scale), weight=float(weight),
            max_version=int(self.max_version), max_staleness=int(max_staleness), opt=0 if adam else 1,
            lr0=float(lr0), decay=float(decay), decay_steps=int(steps), beta1=0.9, beta2=0.999, eps=1e-8,
            momentum=float(r.momentum), weight_decay=float(r.weight_decay), timeout_ticks=self.timeout_ticks,
        )

    def _launch(self, lr: LrLike, mode: int, **kw) -> ContributionResult:
        before = self.version if self.sync_status else None
        self._last_lr = lr
        with self._lock:
            self._launched += 1
            seq = self._launched
            nv.hip().ulog_step(self._args(lr, mode, **kw), self.blocks, self.stream.cuda_stream)
        if self.sync_status:
            self.wait_status()
        else:
            self.check()
        s = self._status_np
        stale = int(s[US_SEQ]) != seq
        last = int(s[US_LAST])
        version = int(s[US_VERSION])
        applied = int(s[US_APPLIED]) if not stale else 0
        if before is not None:
            applied = version - before
        self.replica.updates_applied = version
        return ContributionResult(bool(last & 1) and not stale, (last >> 8) if (last & 1) else -1, applied, version)

    # ---------------------------------------------------------------------------------------------
    def forward_backward(self, replica, abortable: bool = True) -> bool:
        """Enqueues forward + backward. Short-circuiting needs no host decision: every kernel of a
        step program recorded with an abort reference returns at once when the update this replica
        works for is complete. Always True: whether the pass was cut short is the GPU's business."""
        replica.forward_backward()
        return True

    def abort_requested(self) -> bool:
        h = nv.hip()
        return int(h.ulog_peek(self._own_ctl, int(h.UW_ABORT), 1)[0]) > self.version

    def contributors_hint(self) -> int:
        """Gradients in the last update this replica applied (never touches the GPU)."""
        return int(self._status_np[US_CONTRIB])

    def contribute(self, lr: LrLike, *, weight: float = 1.0, quorum: int, abort_at: int = 0, scale: float = 0.0) -> ContributionResult:
        """Offers the gradient for the update that turns this replica's version into the next one;
        on the device: accepted -> wait for the quorum and the owners' shards; stale or late -> drop
        it; then apply every closed update not applied yet."""
        return self._launch(lr, MODE_CONTRIBUTE, weight=weight, quorum=quorum, abort_at=abort_at, scale=scale)

    def skip(self, lr: LrLike) -> ContributionResult:
        """Contributes nothing: waits for the others to close the current update, then applies it."""
        return self._launch(lr, MODE_SKIP)

    def contribute_open(self, lr: LrLike, max_staleness: int = -1) -> ContributionResult:
        """Interval mode: adds the gradient to whatever update is open; waits for nobody."""
        return self._launch(lr, MODE_OPEN, max_staleness=max_staleness)

    def catch_up(self, lr: LrLike, reduce_own: bool = True) -> int:
        return self._launch(lr, MODE_CATCH_UP).applied

    def close_window(self) -> int:
        """Chief's interval timer (its own thread, its own stream): closes the open update if anything
        arrived, reduces it and publishes it. Returns the closed update or -1."""
        if self.close_stream is None:
            with torch.cuda.device(self._device):
                self.close_stream = torch.cuda.Stream()
        args = self._args(0.0, MODE_CATCH_UP, status=self.close_status)
        nv.hip().ulog_close(args, self.blocks, self.close_stream.cuda_stream)
        self.close_stream.synchronize()
        return int(self._close_np[0]) - 1

    def mark_dead(self, rank: int) -> None:
        """The watchdog gave up on `rank`: the kernels stop waiting for it in the back-pressure, and
        the shards it owned in updates it had joined are summed by a stand-in. Called from the
        watchdog's thread; the one-thread kernel runs on a stream of its own, because the training
        stream may be occupied by a launch that is waiting for exactly this."""
        with torch.cuda.device(self._device):
            if self._ctl_stream is None:
                self._ctl_stream = torch.cuda.Stream()
            nv.hip().ulog_mark_dead(self._own_ctl, 1 << rank, self._ctl_stream.cuda_stream)
            self._ctl_stream.synchronize()

    # ---------------------------------------------------------------------------------------------
    def finish(self, final_version: Optional[int] = None, collective: bool = True) -> None:
        """End of training, called by all replicas after the last contribution (and after the chief
        stopped its timer): applies what is still closed-but-unapplied until everybody agrees.
        collective=False (a worker died): one last catch-up, no agreement round."""
        if self._closed:
            return
        self.stream.synchronize()
        if not collective:
            sync, self.sync_status = self.sync_status, True
            try:
                self._launch(self._last_lr, MODE_CATCH_UP)
            finally:
                self.sync_status = sync
            return
        self.comm.barrier()
        lr = self._last_lr
        deadline = time.time() + 60
        while True:
            sync, self.sync_status = self.sync_status, True
            try:
                self._launch(lr, MODE_CATCH_UP)
            finally:
                self.sync_status = sync
            versions = self.comm.all_gather_object(self.version)
            if len(set(versions)) == 1 or time.time() > deadline:
                break

    def close(self) -> None:
        if self._closed:
            return
        self._closed = True
        h = nv.hip()
        torch.cuda.synchronize()
        self.comm.barrier()
        for p in self._imported:
            h.ipc_release(p)
        self._imported.clear()
        self.comm.barrier()
        h.ipc_free(self._own_stage)
        h.ipc_free(self._own_red)
        if not self._ctl_is_replicas:
            h.ipc_free(self._own_ctl)
