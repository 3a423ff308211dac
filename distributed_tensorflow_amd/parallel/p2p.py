"""Hand-written two-shot all-reduce over peer-mapped HBM (csrc/comm/p2p_allreduce.hip).

One GPU per process, one node. Every rank allocates an exchange buffer and a flag block, exports
them as HIP IPC handles, and maps everybody else's; the collective itself is a single kernel
launch per rank on the caller's stream, with no host synchronisation. Per-rank weights turn it
into the masked (backup workers) or weighted (least-squares) aggregation of the reference
(DistributedLeNet/src/backup_worker_optimizer.py:241-270) without a second pass over the data.

RCCL stays the default data plane (Communicator.all_reduce_async); this path is selected with
--allreduce_impl=p2p and is the one that can skip masked-out replicas' buffers entirely.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from ..ops import native as nv


class P2PAllReduce:
    TICKS_PER_SECOND = 100_000_000   # wall_clock64() runs at 100 MHz

    def __init__(self, comm, max_floats: int, blocks: int = 64, timeout_s: float = 10.0):
        nv.require()
        if comm.world > 8:
            raise ValueError("the peer-to-peer all-reduce covers one node (at most 8 GPUs)")
        self.comm = comm
        self.rank, self.world = comm.rank, comm.world
        self.capacity = (max_floats + 3) // 4 * 4
        self.blocks = blocks
        self.timeout_ticks = int(timeout_s * self.TICKS_PER_SECOND)
        self.seq = 0
        h = nv.hip()
        # fine-grained: peers read this buffer while the owner's kernel is still running, so its
        # lines must not sit in a non-coherent cache on either side
        self._own_x = h.ipc_alloc(self.capacity * 4, True)
        self._own_f = h.ipc_alloc(int(h.P2P_FLAG_BLOCK_BYTES), True)
        handles = comm.all_gather_object((h.ipc_export(self._own_x), h.ipc_export(self._own_f)))
        self._xbuf: List[int] = []
        self._flags: List[int] = []
        self._imported: List[int] = []
        error = None
        try:
            for r, (hx, hf) in enumerate(handles):
                if r == self.rank:
                    self._xbuf.append(self._own_x)
                    self._flags.append(self._own_f)
                else:
                    px = h.ipc_import(hx)
                    self._imported.append(px)
                    pf = h.ipc_import(hf)
                    self._imported.append(pf)
                    self._xbuf.append(px)
                    self._flags.append(pf)
        except Exception as e:
            error = f"rank {self.rank}: {e!r}"
        # every rank learns whether all of them could map their peers: nobody is left in a collective
        errors = [e for e in comm.all_gather_object(error) if e]
        if errors:
            raise RuntimeError("peer-to-peer all-reduce: peer HBM cannot be mapped (" + "; ".join(errors) +
                               "); it needs all ranks on one node with peer access - use the RCCL all-reduce")
        self.error = torch.zeros(4, device=comm.device, dtype=torch.int32)
        comm.barrier()

    def all_reduce(self, flat: torch.Tensor, weights: Optional[Sequence[float]] = None) -> torch.Tensor:
        """flat <- sum_r weights[r] * flat_r, in place, on the current stream (every rank must call
        with the same size and weights, in the same order)."""
        n = flat.numel()
        if n > self.capacity or n % 4 or flat.dtype != torch.float32 or not flat.is_cuda or not flat.is_contiguous():
            raise ValueError("p2p all_reduce: needs a contiguous fp32 GPU tensor of at most `max_floats` (multiple of 4) elements")
        w = [1.0] * self.world if weights is None else [float(x) for x in weights]
        if len(w) != self.world:
            raise ValueError("one weight per rank")
        self.seq += 1
        nv.hip().p2p_allreduce(dict(xbuf=self._xbuf, flags=self._flags, weight=w, rank=self.rank, world=self.world, n=n,
                                    seq=self.seq, timeout_ticks=self.timeout_ticks, data=flat.data_ptr(),
                                    error=self.error.data_ptr()), self.blocks, nv.stream())
        return flat

    def check(self) -> None:
        """Synchronises and raises if any wait inside the collective timed out. The error word is
        sticky (every later call returns early with unreduced data), so callers look at it wherever
        they synchronise anyway - the driver does at every loss read-back and before a checkpoint."""
        if int(self.error[0].item()) != 0:
            raise RuntimeError("peer-to-peer all-reduce timed out waiting for a peer; gradients since then were NOT reduced "
                               "and the replicas have diverged - restart from the last checkpoint")

    def close(self) -> None:
        h = nv.hip()
        torch.cuda.synchronize()
        self.comm.barrier()
        for p in self._imported:
            h.ipc_release(p)
        self._imported.clear()
        self.comm.barrier()
        if self._own_x:
            h.ipc_free(self._own_x)
            h.ipc_free(self._own_f)
            self._own_x = self._own_f = 0


def link_probe(comm, payload_bytes: int = 1 << 20, iters: int = 200, cas_iters: int = 2000, timeout_s: float = 10.0) -> dict:
    """First-contact probe between ranks 0 and 1 (csrc/comm/link_probe.hip): ping-pong of one flag
    word, then of the flag plus `payload_bytes`, over IPC-mapped fine-grained HBM with the update
    log's own primitives (release store / acquire load / compare-exchange at system scope), and a
    compare-exchange counter both ranks increment. Every rank of `comm` must call it; ranks above 1
    only take part in the collectives. Returns, on ranks 0 and 1:
      flag_round_trip_us, payload_round_trip_us, payload_gb_s (both directions), bad_words, cas_lost,
      cas_retries, timed_out."""
    nv.require()
    h = nv.hip()
    ticks_per_s = P2PAllReduce.TICKS_PER_SECOND
    floats = max(0, payload_bytes // 16 * 4)
    own_ctl = h.ipc_alloc(int(h.LINK_PROBE_CTL_BYTES), True)
    own_buf = h.ipc_alloc(max(16, floats * 4), True)
    handles = comm.all_gather_object((h.ipc_export(own_ctl), h.ipc_export(own_buf)))
    out: dict = {}
    imported: List[int] = []
    error = None
    try:
        if comm.rank < 2 and comm.world >= 2:
            peer = 1 - comm.rank
            peer_ctl, peer_buf = (h.ipc_import(x) for x in handles[peer])
            imported += [peer_ctl, peer_buf]
    except Exception as e:   # noqa: BLE001 - reported on every rank below
        error = f"rank {comm.rank}: {e!r}"
    errors = [e for e in comm.all_gather_object(error) if e]
    if errors:
        h.ipc_free(own_ctl)
        h.ipc_free(own_buf)
        raise RuntimeError("link probe: peer HBM cannot be mapped (" + "; ".join(errors) + ")")
    comm.barrier()
    if comm.rank < 2 and comm.world >= 2:
        result = torch.zeros(8, dtype=torch.int64).pin_memory()
        word = 16 * 8   # bytes between control words (one per 128-byte line)
        counter = (own_ctl if comm.rank == 0 else peer_ctl) + int(h.LP_COUNTER) * word
        h.link_probe(dict(own_ctl=own_ctl, peer_ctl=peer_ctl, own_buf=own_buf, peer_buf=peer_buf, counter=counter,
                          payload_floats=floats, iters=iters, cas_iters=cas_iters, role=comm.rank,
                          timeout_ticks=int(timeout_s * ticks_per_s), result=result.data_ptr()), nv.stream())
        torch.cuda.synchronize()
        r = result.numpy()
        out = dict(flag_round_trip_us=float(r[0]) / ticks_per_s * 1e6 / iters,
                   payload_round_trip_us=float(r[1]) / ticks_per_s * 1e6 / iters if floats else 0.0,
                   payload_gb_s=(2.0 * floats * 4 * iters) / (float(r[1]) / ticks_per_s) / 1e9 if floats and r[1] else 0.0,
                   bad_words=int(r[2]), timed_out=bool(r[3]), cas_retries=int(r[4]), payload_bytes=floats * 4, iters=iters)
    comm.barrier()
    if comm.rank == 0 and comm.world >= 2:
        final = int(h.ulog_peek(own_ctl, int(h.LP_COUNTER), 1)[0])
        out["cas_lost"] = 2 * cas_iters - final
    lost = comm.all_gather_object(out.get("cas_lost"))[0]
    if out:
        out["cas_lost"] = lost
    torch.cuda.synchronize()
    comm.barrier()
    for p in imported:
        h.ipc_release(p)
    comm.barrier()
    h.ipc_free(own_ctl)
    h.ipc_free(own_buf)
    return out
