"""Process-group bootstrap and the flat-bucket gradient collectives.

One process per GPU. The data plane is RCCL over xGMI (torch.distributed backend "nccl" on
ROCm); on machines without a GPU the same code runs over gloo, which is what the multi-process
CPU tests use. This replaces the gRPC tensor transport between TensorFlow workers and parameter
servers (reference: DistributedLeNet/src/mnist_distributed_train.py:25-35): instead of one RPC
per variable per direction per step, a step moves each gradient bucket exactly once.

xGMI is a point-to-point mesh (7 links per GPU), and the gradients here are small (1-7 MB), so
the exchange is latency bound: one bucket (issued on the compute stream behind backward) for
gradients up to 4 MB, else three large buckets, each issued as soon as the backward pass has
finished its layers and overlapped with the rest of backward (default_grad_buckets below).
"""
from __future__ import annotations

import datetime
import os
from dataclasses import dataclass
from typing import List, Optional

import torch
import torch.distributed as dist


def default_grad_buckets(num_params: int) -> int:
    """Buckets for a gradient of `num_params` fp32 values when nothing else asks for segments.
    Over xGMI an all-reduce of a few MB is latency bound (tens of microseconds), less than what
    splitting it costs: every extra bucket is a segment boundary of the step program (~14 us) plus
    the hand-offs to and from the collective stream. Measured through a one-rank RCCL group,
    ResNet-20 (1.1 MB) at batch 1024: 2.85 ms per step with one bucket issued on the compute
    stream, 2.91 ms with three overlapped ones, 2.83 ms without collectives. Larger gradients get
    three buckets, overlapped with the rest of backward."""
    return 1 if num_params * 4 <= (4 << 20) else 3


def device_identity(device: torch.device) -> str:
    """A string that is equal for two ranks exactly when they sit on the same physical GPU
    (host name + PCI address, or the device UUID where the driver reports one)."""
    import socket

    if device.type != "cuda":
        return "%s/cpu" % socket.gethostname()
    p = torch.cuda.get_device_properties(device)
    pci = (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", 0), getattr(p, "pci_device_id", 0))
    ident = "%04x:%02x:%02x" % pci
    uuid = getattr(p, "uuid", None)
    has_uuid = uuid is not None and str(uuid).strip("0-") != ""
    if has_uuid:
        ident += "/" + str(uuid)
    if not has_uuid and not any(pci):
        # a driver that reports neither: fall back to what the process was told to use, so that the
        # GPUs of a node are never mistaken for one device (ranks that share a GPU for a rehearsal
        # run with the same visibility mask and the same index, and still compare equal)
        masks = ",".join("%s=%s" % (k, os.environ.get(k, "")) for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"))
        ident += "/%s/index%d" % (masks, device.index if device.index is not None else torch.cuda.current_device())
    return "%s/%s" % (socket.gethostname(), ident)


@dataclass
class CommInfo:
    rank: int
    world: int
    local_rank: int
    backend: str
    device: torch.device


class Communicator:
    """Thin owner of the default process group (or a no-op for a single replica)."""

    def __init__(self, info: CommInfo, owns_group: bool, collectives: Optional[bool] = None):
        self.info = info
        self._owns = owns_group
        # False: a single replica, every collective is a no-op. A one-rank process group can be
        # forced on (from_env(force_group=True)) to run the real backend's code path on one GPU.
        self._collectives = info.world > 1 if collectives is None else collectives
        self._pending: List = []
        self.collectives_issued = 0   # data-plane collectives started by this rank (tests count them per step)
        self.p2p = None   # optional hand-written peer-to-peer all-reduce (parallel/p2p.py)

    # -- construction ----------------------------------------------------------------------------
    @classmethod
    def from_env(cls, backend: Optional[str] = None, device: Optional[str] = None, timeout_s: int = 600,
                 force_group: bool = False) -> "Communicator":
        """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT (torchrun contract).
        force_group: create the process group even for one rank, so that the collectives go
        through the backend (rehearsal of the multi-GPU code path on a single GPU)."""
        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
        use_gpu = torch.cuda.is_available() and device != "cpu"
        if backend is None:
            backend = "nccl" if use_gpu else "gloo"
        if use_gpu:
            dev = torch.device("cuda", local_rank % max(1, torch.cuda.device_count()))
            torch.cuda.set_device(dev)
        else:
            dev = torch.device("cpu")
        owns = False
        if world > 1 or force_group:
            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29500")
                kwargs = dict(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s))
                if backend == "nccl":
                    kwargs["device_id"] = dev
                dist.init_process_group(**kwargs)
                owns = True
        return cls(CommInfo(rank, world, local_rank, backend, dev), owns, world > 1 or force_group)

    @classmethod
    def single(cls, device: str | torch.device = "cpu") -> "Communicator":
        return cls(CommInfo(0, 1, 0, "none", torch.device(device)), False)

    # -- properties ------------------------------------------------------------------------------
    @property
    def rank(self) -> int:
        return self.info.rank

    @property
    def world(self) -> int:
        return self.info.world

    @property
    def is_chief(self) -> bool:
        return self.info.rank == 0

    @property
    def device(self) -> torch.device:
        return self.info.device

    # -- collectives -----------------------------------------------------------------------------
    def all_reduce_async(self, flat: torch.Tensor) -> None:
        """SUM all-reduce of one contiguous bucket; returns immediately. On RCCL the collective
        runs on the process group's own stream, ordered after everything already enqueued on the
        current stream, so later kernels on the current stream overlap with it."""
        if not self._collectives:
            return
        self.collectives_issued += 1
        if self.p2p is not None and flat.is_cuda:
            self.p2p.all_reduce(flat)   # stream-ordered kernel on the current stream
            return
        if flat.is_cuda and self.info.backend == "gloo":
            # gloo is the CPU control/test plane: stage device buffers through the host
            torch.cuda.current_stream().synchronize()
            host = flat.detach().cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            flat.copy_(host)
            return
        self._pending.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True))

    def all_reduce_inline(self, flat: torch.Tensor) -> None:
        """SUM all-reduce of a bucket that nothing can overlap with (the last one of a step): a
        blocking-style call, which the RCCL process group issues on the current stream, so the
        step pays no event hand-off to and from the group's own stream."""
        if not self._collectives:
            return
        if self.p2p is not None or not flat.is_cuda or self.info.backend != "nccl":
            self.all_reduce_async(flat)
            return
        self.collectives_issued += 1
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)

    def wait_all(self) -> None:
        """Orders the current stream (or blocks the host, on gloo) after all pending buckets."""
        for w in self._pending:
            w.wait()
        self._pending.clear()

    def all_reduce(self, t: torch.Tensor, op=dist.ReduceOp.SUM) -> torch.Tensor:
        if self._collectives:
            dist.all_reduce(t, op=op)
        return t

    def reduce_scatter_flat(self, flat: torch.Tensor, shard: int) -> torch.Tensor:
        """ONE reduce-scatter over `world` equal contiguous shards of `shard` floats: rank r ends up
        with the SUM of flat[r*shard:(r+1)*shard] in place (returned as a view of `flat`, which is
        the arena's padded buffer, ParamArena.padded)."""
        lo, hi = self.rank * shard, (self.rank + 1) * shard
        if flat.numel() != shard * self.world:
            raise ValueError("reduce_scatter_flat: the buffer must hold exactly `world` shards")
        if not self._collectives:
            return flat[lo:hi]
        self.collectives_issued += 1
        if flat.is_cuda and self.info.backend == "gloo":
            torch.cuda.current_stream().synchronize()
            host = flat.detach().cpu()
            out = torch.empty(shard, dtype=flat.dtype)
            dist.reduce_scatter_tensor(out, host, op=dist.ReduceOp.SUM)
            flat[lo:hi].copy_(out)
            return flat[lo:hi]
        if self.info.backend == "nccl":
            dist.reduce_scatter_tensor(flat[lo:hi], flat, op=dist.ReduceOp.SUM)   # in place: output = own chunk of the input
        else:
            out = torch.empty(shard, dtype=flat.dtype, device=flat.device)
            dist.reduce_scatter_tensor(out, flat, op=dist.ReduceOp.SUM)
            flat[lo:hi].copy_(out)
        return flat[lo:hi]

    def all_gather_flat(self, flat: torch.Tensor, shard: int) -> None:
        """ONE all-gather: every rank contributes flat[rank*shard:(rank+1)*shard] and receives all
        shards in place."""
        lo, hi = self.rank * shard, (self.rank + 1) * shard
        if flat.numel() != shard * self.world:
            raise ValueError("all_gather_flat: the buffer must hold exactly `world` shards")
        if not self._collectives:
            return
        self.collectives_issued += 1
        if flat.is_cuda and self.info.backend == "gloo":
            torch.cuda.current_stream().synchronize()
            host = flat.detach().cpu()
            dist.all_gather_into_tensor(host, host[lo:hi].clone())
            flat.copy_(host)
            return
        if self.info.backend == "nccl":
            dist.all_gather_into_tensor(flat, flat[lo:hi])   # in place: input = own chunk of the output
        else:
            dist.all_gather_into_tensor(flat, flat[lo:hi].clone())

    def broadcast(self, t: torch.Tensor, src: int = 0) -> torch.Tensor:
        if self._collectives:
            if t.is_cuda and self.info.backend == "gloo":
                torch.cuda.current_stream().synchronize()
                host = t.detach().cpu()
                dist.broadcast(host, src=src)
                t.copy_(host)
            else:
                dist.broadcast(t, src=src)
        return t

    def barrier(self) -> None:
        if self._collectives:
            if self.info.backend == "nccl":
                dist.barrier(device_ids=[self.info.device.index])
            else:
                dist.barrier()

    def all_gather_object(self, obj):
        if not self._collectives:
            return [obj]
        out = [None] * self.world
        dist.all_gather_object(out, obj)
        return out

    def topology(self) -> dict:
        """What the ranks of this group really run on: backend, number of ranks, number of
        DISTINCT physical GPUs (ranks rehearsing on a shared GPU count once), and how many ranks
        exchange data through RCCL."""
        ids = self.all_gather_object(device_identity(self.info.device))
        on_gpu = self.info.device.type == "cuda"
        return {
            "backend": self.info.backend if self._collectives else "none",
            "ranks": self.world,
            "distinct_gpus": len(set(ids)) if on_gpu else 0,
            "rccl_ranks": self.world if (self._collectives and self.info.backend == "nccl") else 0,
        }

    def check_health(self) -> None:
        """Raises if a hand-written collective reported a timeout since the last call. Reads a device
        word, so call it where the stream is synchronised anyway (loss read-back, checkpoints)."""
        if self.p2p is not None:
            self.p2p.check()

    def enable_p2p(self, max_floats: int, **kw) -> None:
        """Routes all_reduce_async of GPU buffers through the peer-to-peer kernel."""
        from .p2p import P2PAllReduce

        if self.world > 1 and self.info.device.type == "cuda":
            self.p2p = P2PAllReduce(self, max_floats, **kw)

    def close(self) -> None:
        if self.p2p is not None:
            self.p2p.close()
            self.p2p = None
        if self._owns and dist.is_initialized():
            dist.destroy_process_group()
            self._owns = False
