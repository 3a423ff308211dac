"""Flat parameter arenas.

Every trainable tensor of a model lives at a fixed, 256-byte-aligned offset of ONE contiguous
fp32 buffer; gradients and optimizer slots use identically laid out buffers. This replaces
TensorFlow's `replica_device_setter` variable placement on parameter servers
(reference: DistributedLeNet/src/distributed_train.py:129-132): parameters are replicated (or
sharded by byte range) in each GPU's HBM, a gradient exchange is a single collective over one
buffer, the optimizer is one kernel launch, and the (name -> offset, shape) table doubles as the
checkpoint's variable-name -> tensor map.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Iterable, Iterator, List, Sequence, Tuple

import torch

ALIGN_FLOATS = 64  # 256 bytes: keeps every tensor 16-byte aligned for dwordx4 access
MAX_SHARD_WORLD = 64
# Unused tail behind the last tensor of every flat buffer. Sharded modes cut the buffer into `world`
# EQUAL ranges of ceil(total / (64 * world)) * 64 floats, the last of which reaches into this tail,
# so a reduce-scatter / all-gather is one collective of uniform chunks whatever the model size.
SHARD_PAD_FLOATS = ALIGN_FLOATS * MAX_SHARD_WORLD


@dataclass(frozen=True)
class ParamSpec:
    name: str                 # canonical (scoped) name, e.g. "conv1_0/conv2_in_block/conv"
    shape: Tuple[int, ...]
    init: str                 # "trunc_normal", "zeros", "ones", "const:<v>"
    tf_name: str = ""         # TensorFlow auto-generated name ("Variable_12")
    trainable: bool = True

    @property
    def numel(self) -> int:
        n = 1
        for s in self.shape:
            n *= s
        return n


class ParamArena:
    """name -> view table over flat `params`, `grads` and named slot buffers."""

    def __init__(self, specs: Sequence[ParamSpec], device: torch.device | str = "cpu", slots: Iterable[str] = ()):
        self.specs: List[ParamSpec] = list(specs)
        self.device = torch.device(device)
        self.offsets: Dict[str, int] = {}
        off = 0
        for s in self.specs:
            if s.name in self.offsets:
                raise ValueError(f"duplicate parameter name {s.name}")
            self.offsets[s.name] = off
            off += (s.numel + ALIGN_FLOATS - 1) // ALIGN_FLOATS * ALIGN_FLOATS
        self.total = off
        self.capacity = self.total + SHARD_PAD_FLOATS
        # Every flat buffer has the same layout: [0, total) tensors, [total, capacity) zero padding
        # that only the sharded collectives and the per-shard optimizer ever touch (zeros stay
        # zeros under Adam and momentum SGD).
        self._full: Dict[str, torch.Tensor] = {"params": torch.zeros(self.capacity, dtype=torch.float32, device=self.device)}
        self.params = self._full["params"][: self.total]
        # The gradient buffer carries a small tail behind the padding that the optimizer never
        # touches: strategies put per-step scalars there (e.g. "my gradient is part of this
        # update") so they travel in the same collective as the gradients.
        self.grads_full = torch.zeros(self.capacity + ALIGN_FLOATS, dtype=torch.float32, device=self.device)
        self._full["grads"] = self.grads_full[: self.capacity]
        self.grads = self.grads_full[: self.total]
        self.aux = self.grads_full[self.capacity:]
        self.slots: Dict[str, torch.Tensor] = {}
        for k in slots:
            self._full["slot:" + k] = torch.zeros(self.capacity, dtype=torch.float32, device=self.device)
            self.slots[k] = self._full["slot:" + k][: self.total]
        self._by_name = {s.name: s for s in self.specs}

    # -- views -----------------------------------------------------------------------------
    def spec(self, name: str) -> ParamSpec:
        return self._by_name[name]

    def _view(self, buf: torch.Tensor, name: str) -> torch.Tensor:
        s = self._by_name[name]
        off = self.offsets[name]
        return buf[off:off + s.numel].view(s.shape)

    def param(self, name: str) -> torch.Tensor:
        return self._view(self.params, name)

    def grad(self, name: str) -> torch.Tensor:
        return self._view(self.grads, name)

    def slot(self, slot: str, name: str) -> torch.Tensor:
        return self._view(self.slots[slot], name)

    def names(self) -> List[str]:
        return [s.name for s in self.specs]

    def items(self) -> Iterator[Tuple[str, torch.Tensor]]:
        for s in self.specs:
            yield s.name, self.param(s.name)

    @property
    def num_parameters(self) -> int:
        return sum(s.numel for s in self.specs)

    # -- init --------------------------------------------------------------------------------
    def initialize(self, seed: int = 66478, stddev: float = 0.1) -> None:
        """Initialises every tensor per its spec. Weights are truncated-normal(0, stddev) cut
        at two standard deviations like tf.truncated_normal (reference:
        DistributedLeNet/src/mnist.py:81-101, DistributedResNet/src/resnet.py:41-46); the
        generator is seeded per model so all ranks start from identical parameters."""
        gen = torch.Generator(device="cpu")
        gen.manual_seed(seed)
        host = torch.zeros(self.total, dtype=torch.float32)
        for s in self.specs:
            off = self.offsets[s.name]
            view = host[off:off + s.numel]
            if s.init == "trunc_normal":
                torch.nn.init.trunc_normal_(view, mean=0.0, std=stddev, a=-2 * stddev, b=2 * stddev, generator=gen)
            elif s.init == "zeros":
                view.zero_()
            elif s.init == "ones":
                view.fill_(1.0)
            elif s.init.startswith("const:"):
                view.fill_(float(s.init.split(":", 1)[1]))
            else:
                raise ValueError(f"unknown init {s.init}")
        self.params.copy_(host)
        self.grads.zero_()
        for t in self.slots.values():
            t.zero_()

    # -- sharding (reduce-scatter / async parameter shards) -----------------------------------------
    def shard_size(self, world: int) -> int:
        if not 1 <= world <= MAX_SHARD_WORLD:
            raise ValueError(f"sharding supports 1..{MAX_SHARD_WORLD} ranks")
        quanta = (self.total + ALIGN_FLOATS - 1) // ALIGN_FLOATS
        return (quanta + world - 1) // world * ALIGN_FLOATS

    def shard_bounds(self, world: int) -> List[Tuple[int, int]]:
        """`world` EQUAL contiguous ranges of `shard_size(world)` floats covering [0, total); the
        last ranges may reach into the zero padding behind `total` (index the buffers returned
        by `padded()` with them, not `params` / `grads`)."""
        n = self.shard_size(world)
        return [(r * n, (r + 1) * n) for r in range(world)]

    def full(self, which: str) -> torch.Tensor:
        """A flat buffer with its padding ("params", "grads", "slot:<name>"): shard ranges from
        `shard_bounds` index this, since the last shards reach behind `total`."""
        return self._full[which]

    def padded(self, which: str, world: int) -> torch.Tensor:
        """View of a flat buffer ("params", "grads", "slot:m", ...) over all `world` equal shards,
        padding included: what the sharded collectives and the per-shard optimizer address."""
        return self._full[which][: self.shard_size(world) * world]

    def state_dict(self, use_tf_names: bool = False) -> Dict[str, torch.Tensor]:
        out = {}
        for s in self.specs:
            key = s.tf_name if (use_tf_names and s.tf_name) else s.name
            out[key] = self.param(s.name).detach().cpu().clone()
        return out

    def load_state_dict(self, state: Dict[str, torch.Tensor], strict: bool = True) -> List[str]:
        """Loads by canonical or TensorFlow name; returns the names that were found."""
        found = []
        for s in self.specs:
            src = state.get(s.name)
            if src is None and s.tf_name:
                src = state.get(s.tf_name)
            if src is None:
                if strict:
                    raise KeyError(f"missing tensor for {s.name} / {s.tf_name}")
                continue
            if tuple(src.shape) != tuple(s.shape):
                raise ValueError(f"shape mismatch for {s.name}: {tuple(src.shape)} vs {s.shape}")
            self.param(s.name).copy_(src.to(self.device, torch.float32))
            found.append(s.name)
        return found
