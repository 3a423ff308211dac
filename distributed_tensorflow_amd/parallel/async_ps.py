"""Asynchronous SGD over one-sided, staleness-bounded parameter shards.

The reference only sketches asynchronous training (reference: backup_worker_optimizer.py:44-51
docstring; `num_replicas_to_aggregate=1` in cfg/50_workers/50_workers_1_aggregate_sync:10) on top
of TensorFlow's ConditionalAccumulator, whose rule is: a gradient tagged with a local step older
than the accumulator's global step is dropped. Here that rule becomes a first-class mode without a
parameter-server process:

  * the flat parameter arena is cut into `world` contiguous shards; replica r OWNS shard r: the
    authoritative parameters and the Adam slots of that range live in r's memory (HBM on a GPU);
  * every replica maps every owner's shard and a per-sender gradient inbox (CUDA IPC over xGMI
    between GPUs of one node, a shared file mapping between CPU processes);
  * pull = one-sided read of all shards; push = one-sided write of each gradient slice into the
    owner's inbox followed by a doorbell (local step of the pull) on the shared-memory board;
  * the owner applies a pushed gradient immediately iff `shard_version - pushed_step <= tau`
    (staleness bound; tau = 0 is the accumulator's "drop anything stale" rule) and bumps the shard
    version; otherwise the gradient is dropped. No replica ever waits for another replica's
    compute: waiting only happens when an inbox slot is still unread, bounded by a timeout.
"""
from __future__ import annotations

import os
import time
from dataclasses import dataclass
from typing import List, Tuple

import torch

from ..ops import reference as ref
from .comm import Communicator


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


@dataclass
class AsyncStats:
    pushed: int = 0
    applied: int = 0
    dropped_stale: int = 0
    push_waits: int = 0            # host: polls spent waiting for an inbox slot; device: pushes given up on a busy slot


class AsyncParameterShards:
    """Shard ownership, peer mappings and the push / pull / apply protocol. This class is the host
    implementation (CPU replicas; data in shared file mappings, doorbells on the board); for GPU
    replicas the constructor hands back `async_ps_hip.DeviceAsyncShards`, the same protocol run by
    kernels over peer-mapped HBM."""

    def __new__(cls, replica, *args, **kw):
        if cls is AsyncParameterShards and replica.arena.params.is_cuda:
            from .async_ps_hip import DeviceAsyncShards

            return object.__new__(DeviceAsyncShards)
        return object.__new__(cls)

    def __init__(self, replica, comm: Communicator, board, staleness_bound: int, name: str, lr_fn,
                 push_timeout_s: float = 30.0, start_version: int = 0):
        if board is None:
            raise ValueError("asynchronous shards need the shared-memory control board")
        self.replica, self.comm, self.board = replica, comm, board
        self.tau = max(0, int(staleness_bound))
        self.rank, self.world = comm.rank, comm.world
        self.arena = replica.arena
        self.bounds = self.arena.shard_bounds(self.world)
        self.lr_fn = lr_fn
        self.push_timeout_s = push_timeout_s
        self.stats = AsyncStats()
        self.staleness_hist: List[int] = [0] * 16
        self.on_gpu = False
        self.start_version = max(0, int(start_version))
        self._name = name.strip("/").replace("/", "_")
        self._pulled_version: List[int] = [self.start_version] * self.world
        self._files: List[str] = []
        self._slot_names = sorted(self.arena.slots)          # "m", "v" (Adam) or "m" (momentum SGD)
        # a resumed run continues the restored step count: learning-rate staircase and Adam's bias
        # correction are functions of the shard's version
        self.board.set_shard_version(self.rank, self.start_version)
        self._map_regions()

    # ------------------------------------------------------------------------------------------
    def _map_regions(self) -> None:
        """The owner creates its regions - the authoritative parameters of its shard, its optimizer
        slots and one inbox per sender - from its (restored or freshly initialised) arena, and
        everybody maps all of them. The slots are mapped by everybody so that the chief can copy
        the whole optimizer state into its arena for a checkpoint (`gather_state`)."""
        lo, hi = self.bounds[self.rank]
        n_own = hi - lo
        for s in range(self.world):
            self._files.append(f"{self._name}_inbox_{self.rank}_{s}")
            _shm_tensor(self._files[-1], n_own, True)
        self._files.append(f"{self._name}_shard_{self.rank}")
        own = _shm_tensor(self._files[-1], n_own, True)
        own.copy_(self.arena.full("params")[lo:hi])
        for k in self._slot_names:
            self._files.append(f"{self._name}_slot_{k}_{self.rank}")
            _shm_tensor(self._files[-1], n_own, True).copy_(self.arena.full("slot:" + k)[lo:hi])
        self.comm.barrier()
        self.shards = []
        self.inboxes: List[List[torch.Tensor]] = []
        self.slot_regions: List[dict] = []
        for o in range(self.world):
            n_o = self.bounds[o][1] - self.bounds[o][0]
            self.shards.append(_shm_tensor(f"{self._name}_shard_{o}", n_o, False))
            self.inboxes.append([_shm_tensor(f"{self._name}_inbox_{o}_{s}", n_o, False) for s in range(self.world)])
            self.slot_regions.append({k: _shm_tensor(f"{self._name}_slot_{k}_{o}", n_o, False) for k in self._slot_names})
        self.comm.barrier()

    # ------------------------------------------------------------------------------------------
    def gather_state(self) -> None:
        """Copies every owner's authoritative parameters and optimizer slots into the LOCAL arena
        (one-sided reads of the mapped regions; nobody is stopped, so the cut has the consistency
        of a pull: each shard at some version at or after the one its owner last published).
        The chief calls it before a checkpoint so that the saved state holds all variables, the
        optimizer's included (reference: DistributedLeNet/src/distributed_train.py:212,234-242,370-373)."""
        a = self.arena
        for o, (lo, hi) in enumerate(self.bounds):
            if hi <= lo:
                continue
            a.full("params")[lo:hi].copy_(self.shards[o][: hi - lo])
            for k in self._slot_names:
                a.full("slot:" + k)[lo:hi].copy_(self.slot_regions[o][k][: hi - lo])

    # ------------------------------------------------------------------------------------------
    def pull(self) -> None:
        """One-sided read of every shard into the local parameter arena."""
        p = self.arena.full("params")
        with self.replica.stream_ctx():
            for o, (lo, hi) in enumerate(self.bounds):
                self._pulled_version[o] = self.board.shard_version(o)
                if hi > lo:
                    p[lo:hi].copy_(self.shards[o][: hi - lo], non_blocking=True)
        self.replica.synchronize()

    def push(self) -> None:
        """One-sided write of each gradient slice into its owner's inbox + doorbell."""
        g = self.arena.full("grads")
        for k in range(self.world):
            o = (self.rank + k) % self.world  # start with the own shard, spread the peers
            lo, hi = self.bounds[o]
            deadline = time.time() + self.push_timeout_s
            while self.board.poll_doorbell(o, self.rank) >= 0:
                # the owner has not consumed the previous push yet
                self.stats.push_waits += 1
                self.service()
                if time.time() > deadline or self.board.should_stop():
                    break
                time.sleep(0.0002)
            if self.board.poll_doorbell(o, self.rank) >= 0:
                continue  # still busy: this gradient slice is dropped
            if hi > lo:
                with self.replica.stream_ctx():
                    self.inboxes[o][self.rank][: hi - lo].copy_(g[lo:hi], non_blocking=True)
                self.replica.synchronize()
            self.board.ring_doorbell(o, self.rank, self._pulled_version[o])
            self.stats.pushed += 1

    def service(self) -> int:
        """Applies (or drops) every gradient waiting in this replica's inbox; returns #applied."""
        me = self.rank
        lo, hi = self.bounds[me]
        n = hi - lo
        applied = 0
        for s in range(self.world):
            pushed_step = self.board.poll_doorbell(me, s)
            if pushed_step < 0:
                continue
            version = self.board.shard_version(me)
            self.staleness_hist[min(15, max(0, version - pushed_step))] += 1
            if version - pushed_step <= self.tau:
                if n > 0:
                    self._apply(self.inboxes[me][s][:n], version)
                self.board.bump_shard_version(me)
                applied += 1
                self.stats.applied += 1
                if me == 0:
                    self.board.set_global_step(self.board.shard_version(0))
            else:
                self.stats.dropped_stale += 1
            self.board.clear_doorbell(me, s)
        return applied

    def _apply(self, grad: torch.Tensor, version: int) -> None:
        lo, hi = self.bounds[self.rank]
        lr = self.lr_fn(version) if callable(self.lr_fn) else float(self.lr_fn)
        shard = self.shards[self.rank][: hi - lo]
        slots = self.slot_regions[self.rank]
        if self.replica.optimizer == "adam":
            ref.adam_update(shard, grad, slots["m"][: hi - lo], slots["v"][: hi - lo], ref.adam_lr_t(lr, version + 1))
        else:
            ref.momentum_update(shard, grad, slots["m"][: hi - lo], lr, self.replica.momentum, self.replica.weight_decay)

    # ------------------------------------------------------------------------------------------
    def step(self) -> Tuple[float, int]:
        """One asynchronous worker iteration; returns (loss, global_step)."""
        self.service()
        self.pull()
        self.replica.forward_backward()
        self.replica.synchronize()
        self.push()
        self.service()
        return self.replica.loss_value(), self.board.global_step()

    def own_state(self) -> dict:
        """Copies of the own shard's authoritative parameters and optimizer slots (tests, debugging)."""
        out = {"params": self.shards[self.rank].clone()}
        out.update({k: t.clone() for k, t in self.slot_regions[self.rank].items()})
        return out

    def versions(self) -> List[int]:
        """Updates applied to every shard so far."""
        return [int(self.board.shard_version(o)) for o in range(self.world)]

    def drain(self, timeout_s: float = 30.0) -> None:
        """End of training: keep servicing the own shard until every replica has stopped pushing
        (a finished owner must not leave the others' last pushes unread)."""
        done_mark = -7
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
        self.gather_state()      # final parameters AND every owner's optimizer slots, for the last checkpoint

    def close(self) -> None:
        for f in self._files:
            try:
                os.unlink(os.path.join("/dev/shm", f))
            except OSError:
                pass
        self._files = []
