"""Asynchronous parameter shards in HBM (csrc/comm/async_shards.hip): the device implementation of
`parallel/async_ps.py`.

Same ownership and staleness rule, but nothing on the training path touches the host: pull, push
and the owner's apply are kernels, the optimizer hyper-parameters are derived on the device from
the shard's version, and a service thread of the owner launches a short sweep kernel on a side
stream every `service_period_s`, which applies the gradients waiting in the shard's inboxes WHILE
the owner's own forward / backward pass runs - and whatever the owner's training thread is doing.
The training thread only enqueues; counters (applied / dropped / staleness histogram, the versions seen by the last pull)
are mirrored into pinned memory by the kernels.
"""
from __future__ import annotations

import os
import threading
import time
from typing import List, Tuple

import torch

from ..lr import schedule_params
from ..ops import native as nv
from .async_ps import AsyncParameterShards, AsyncStats


class DeviceAsyncShards(AsyncParameterShards):
    TICKS_PER_SECOND = 100_000_000

    def __init__(self, replica, comm, board, staleness_bound: int, name: str, lr_fn, push_timeout_s: float = 30.0,
                 start_version: int = 0, blocks: int = 32, service_blocks: int = 16, service_period_s: float = 0.0005):
        nv.require()
        h = nv.hip()
        if board is None:
            raise ValueError("asynchronous shards need the shared-memory control board")
        if comm.world > 8:
            raise ValueError("device-resident asynchronous shards cover one node (at most 8 GPUs)")
        self.replica, self.comm, self.board = replica, comm, board
        self.tau = max(0, int(staleness_bound))
        self.rank, self.world = comm.rank, comm.world
        self.arena = replica.arena
        self.bounds = self.arena.shard_bounds(self.world)
        self.shard_floats = self.arena.shard_size(self.world)
        self.lr_fn = lr_fn
        self.stats = AsyncStats()
        self.staleness_hist: List[int] = [0] * int(h.ASYNC_HIST_BINS)
        self.on_gpu = True
        self.blocks = int(os.environ.get("DTF_ASYNC_BLOCKS", blocks))
        self.service_blocks = int(os.environ.get("DTF_ASYNC_SERVICE_BLOCKS", service_blocks))
        self.read_loss_every_step = True
        self.stream = replica.stream
        with torch.cuda.device(replica.device):
            self.side = torch.cuda.Stream()
        self._closed = False
        self.service_period_s = service_period_s
        self._svc_lock = threading.Lock()
        self._svc_stop = threading.Event()
        self._svc_thread = None
        self._svc_error = None          # what ended the service thread, raised by the training thread
        self.sweeps = 0
        self._fault_after = int(os.environ.get("DTF_FAULT_ASYNC_SERVICE", "0"))   # tests: the service fails after this many sweeps
        self.start_version = max(0, int(start_version))
        try:
            self._lr = schedule_params(lr_fn)
        except TypeError:
            # an arbitrary callable: sample it (the drivers pass the staircase schedule or a constant)
            self._lr = (float(lr_fn(0)), 1, 1.0)
        n = self.shard_floats
        adam = replica.optimizer == "adam"
        self._slot_names = ["m", "v"] if adam else ["m"]
        # control block, authoritative parameters, inboxes, then one buffer per optimizer slot: all of
        # them mapped by every peer (the slots so that the chief can checkpoint the whole optimizer state)
        self._own = [int(h.ipc_alloc(int(h.ASYNC_CTL_BYTES), True)), int(h.ipc_alloc(n * 4, True)),
                     int(h.ipc_alloc(n * 4 * self.world, True))] + [int(h.ipc_alloc(n * 4, True)) for _ in self._slot_names]
        # the authoritative copy of the own shard and its slots start from the (restored and
        # broadcast) arena; the version from the restored step
        lo = self.rank * n
        a = self.arena
        for ptr, which in zip(self._own[1:2] + self._own[3:], ["params"] + ["slot:" + k for k in self._slot_names]):
            h.memcpy_d2d(ptr, a.padded(which, self.world)[lo:lo + n].data_ptr(), n * 4, self.stream.cuda_stream)
        self.stream.synchronize()
        h.ulog_poke(self._own[0], int(h.AW_VERSION), self.start_version)
        handles = comm.all_gather_object(tuple(h.ipc_export(p) for p in self._own))
        self._ctl, self._shard, self._inbox, self._imported = [], [], [], []
        self._slots = [[] for _ in self._slot_names]
        # a rank that cannot map a peer's HBM (different nodes, no peer access) must not leave the
        # others waiting in the next collective: the ranks exchange the outcome and fail together
        error = None
        try:
            for r, hs in enumerate(handles):
                if r == self.rank:
                    ptrs = list(self._own)
                else:
                    ptrs = []
                    for x in hs:
                        ptrs.append(int(h.ipc_import(x)))
                        self._imported.append(ptrs[-1])
                self._ctl.append(ptrs[0])
                self._shard.append(ptrs[1])
                self._inbox.append(ptrs[2])
                for k in range(len(self._slot_names)):
                    self._slots[k].append(ptrs[3 + k])
        except Exception as e:
            error = f"rank {self.rank}: {e!r}"
        errors = [e for e in comm.all_gather_object(error) if e]
        if errors:
            raise RuntimeError("asynchronous shards: peer-mapped HBM is unavailable (" + "; ".join(errors) +
                               "); the device shards need all ranks on one node with peer access")
        self.status = torch.zeros(40, dtype=torch.int64).pin_memory()
        self._st = self.status.numpy()
        timeout = int(min(push_timeout_s, 20.0) * self.TICKS_PER_SECOND)
        self._base = dict(
            ctl=self._ctl, shard=self._shard, inbox=self._inbox, params=a.padded("params", self.world).data_ptr(),
            grads=a.padded("grads", self.world).data_ptr(), m=self._slots[0][self.rank],
            v=self._slots[1][self.rank] if adam else None, status=self.status.data_ptr(),
            shard_floats=n, rank=self.rank, world=self.world, tau=self.tau, opt=0 if adam else 1,
            lr0=self._lr[0], decay_steps=self._lr[1], decay=self._lr[2], beta1=0.9, beta2=0.999, eps=1e-8,
            momentum=float(replica.momentum), weight_decay=float(replica.weight_decay), timeout_ticks=timeout,
            push_wait_ticks=int(0.05 * self.TICKS_PER_SECOND),
        )
        comm.barrier()
        self._start_service_thread()

    # ------------------------------------------------------------------------------------------
    def _args(self, **kw) -> dict:
        d = dict(self._base)
        d.update(kw)
        return d

    def _refresh_stats(self) -> None:
        s = self._st
        self.stats.applied, self.stats.dropped_stale = int(s[8]), int(s[9])
        self.stats.pushed, self.stats.push_waits = int(s[10]), int(s[11])
        self.staleness_hist = [int(x) for x in s[12:12 + len(self.staleness_hist)]]

    def pull(self) -> None:
        nv.hip().async_pull(self._args(), self.blocks, self.stream.cuda_stream)

    def push(self) -> None:
        nv.hip().async_push(self._args(), self.blocks, self.stream.cuda_stream)

    def _start_service_thread(self) -> None:
        dev = self.replica.device

        def loop() -> None:
            torch.cuda.set_device(dev)
            while not self._svc_stop.wait(self.service_period_s):
                try:
                    self.service()
                except Exception as e:   # noqa: BLE001 - kept for the training thread, which raises it at its next step
                    self._svc_error = e
                    return

        self._svc_thread = threading.Thread(target=loop, name="async-shard-service", daemon=True)
        self._svc_thread.start()

    def _stop_service_thread(self) -> None:
        if self._svc_thread is not None:
            self._svc_stop.set()
            self._svc_thread.join(timeout=30)
            self._svc_thread = None

    def service(self) -> int:
        """One sweep over the own shard's doorbells on the side stream (the service thread's job; the
        end-of-training drain calls it too). Waits for the sweep, so at most one is ever in flight."""
        if self._fault_after and self.sweeps >= self._fault_after:
            raise RuntimeError("injected fault (DTF_FAULT_ASYNC_SERVICE)")
        with self._svc_lock:
            nv.hip().async_service(self._args(), self.service_blocks, self.side.cuda_stream)
            self.side.synchronize()
            self.sweeps += 1
        return 0

    def step(self) -> Tuple[float, int]:
        """One asynchronous worker iteration, enqueued without waiting for the GPU; returns the loss
        (read back unless `read_loss_every_step` is off) and the global step seen by the last pull."""
        h = nv.hip()
        self.check_service()
        self.pull()
        self.replica.forward_backward()
        self.push()
        loss = self.replica.loss_value() if self.read_loss_every_step else float("nan")
        self._refresh_stats()
        self._steps = getattr(self, "_steps", 0) + 1
        if self._steps % 64 == 0 and int(h.ulog_peek(self._own[0], int(h.AW_ERROR), 1)[0]):
            self.board.report_failure(self.rank)
            raise TimeoutError(f"asynchronous shards (rank {self.rank}): a launch's blocks did not all start")
        gs = int(self._st[0])
        if gs > self.board.global_step():
            self.board.set_global_step(gs)
        return loss, max(gs, int(self.board.global_step()))

    def check_service(self) -> None:
        """Every step of the training thread: if the service thread has died, nobody applies the
        gradients pushed to this shard any more (pushers would silently drop every slice), so the
        job is stopped and the error raised here, by name."""
        err = self._svc_error
        if err is None and self._svc_thread is not None and not self._svc_thread.is_alive() and not self._svc_stop.is_set():
            err = RuntimeError("thread ended without an error")
        if err is not None:
            self._svc_error = None
            self.board.report_failure(self.rank)
            raise RuntimeError(f"asynchronous shards (rank {self.rank}): the owner's service thread failed: {err!r}") from err

    def gather_state(self) -> None:
        """Copies every owner's authoritative parameters and optimizer slots into the LOCAL arena
        with device copies from the peer mappings, on the training stream (nobody is stopped; the
        cut has the consistency of a pull). Called by the chief before a checkpoint and by
        everybody at the end of `drain`."""
        h = nv.hip()
        n = self.shard_floats
        a = self.arena
        srcs = [("params", self._shard)] + [("slot:" + k, self._slots[i]) for i, k in enumerate(self._slot_names)]
        for which, ptrs in srcs:
            base = a.padded(which, self.world).data_ptr()
            for o in range(self.world):
                h.memcpy_d2d(base + o * n * 4, ptrs[o], n * 4, self.stream.cuda_stream)
        self.stream.synchronize()

    def own_state(self) -> dict:
        """Copies of the own shard's authoritative parameters and optimizer slots (tests, debugging)."""
        h = nv.hip()
        n = self.shard_floats
        out = {}
        for name, ptr in [("params", self._own[1])] + [(k, self._own[3 + i]) for i, k in enumerate(self._slot_names)]:
            t = torch.empty(n, dtype=torch.float32, device=self.replica.device)
            h.memcpy_d2d(t.data_ptr(), ptr, n * 4, self.stream.cuda_stream)
            out[name] = t
        self.stream.synchronize()
        return out

    def versions(self) -> List[int]:
        h = nv.hip()
        return [int(h.ulog_peek(c, int(h.AW_VERSION), 1)[0]) for c in self._ctl]

    def drain(self, timeout_s: float = 30.0) -> None:
        """End of training: keep servicing the own shard until every replica has stopped pushing,
        then take the final parameters."""
        done_mark = -7
        self.replica.synchronize()
        self._stop_service_thread()
        self.board.set_local_step(self.rank, done_mark)
        deadline = time.time() + timeout_s
        while time.time() < deadline:
            self.service()
            if all(self.board.local_step(r) == done_mark for r in range(self.world)):
                break
            time.sleep(0.0005)
        self.board.barrier(timeout_s)
        self.service()
        self.board.barrier(timeout_s)
        self.pull()              # records the final versions
        self.gather_state()      # final parameters AND every owner's optimizer slots, for the last checkpoint
        self.replica.synchronize()
        self._refresh_stats()
        if self.rank == 0:
            self.board.set_global_step(max(int(self.board.global_step()), self.versions()[0]))

    def close(self) -> None:
        if self._closed:
            return
        self._closed = True
        self._stop_service_thread()
        h = nv.hip()
        torch.cuda.synchronize()
        self.comm.barrier()
        for p in self._imported:
            h.ipc_release(p)
        self._imported.clear()
        self.comm.barrier()
        for p in self._own:
            h.ipc_free(p)
