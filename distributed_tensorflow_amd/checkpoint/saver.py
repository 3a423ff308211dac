"""TensorFlow-compatible checkpoints: variable-name -> tensor bundles plus the `checkpoint` state
file.

What `tf.train.Saver()` over all global variables writes in the reference (reference:
DistributedLeNet/src/distributed_train.py:212,234-242,370-373) is reproduced name for name:
  global_step                       int32 scalar
  Variable, Variable_1, ...         model tensors in creation order (TF auto names), HWIO / [in,out]
  <var>/Adam, <var>/Adam_1          Adam first / second moment slots
  beta1_power, beta2_power          Adam's running beta powers (beta^(t+1) after t updates)
  <var>/Momentum                    momentum-SGD slot (when that optimizer is used)
`names="scoped"` writes the scoped names of the reference's alternative ResNet model file instead
(conv0/conv, conv1_0/conv1_in_block/beta, ..., fc/fc_bias; reference: DistributedResNet/src/back_up.py).
Tensors keep TensorFlow's layouts in memory, so they are written without any transpose. The
bundle files themselves are produced by the native writer (csrc/ckpt/tensor_bundle.cc).
TensorFlow is not installed in this environment, so byte-level parity with a real TF-written
bundle is unpinned; the format is covered by golden-byte tests built from the format definition.
"""
from __future__ import annotations

import glob
import os
import re
import threading
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .. import _host
from ..runtime.arena import ParamArena
from .meta_graph import write_meta

CKPT_BASENAME = "model.ckpt"
_NP_DTYPES = {_host.DT_FLOAT: np.float32, _host.DT_DOUBLE: np.float64, _host.DT_INT32: np.int32, _host.DT_INT64: np.int64}
_DT_OF = {np.dtype(np.float32): _host.DT_FLOAT, np.dtype(np.float64): _host.DT_DOUBLE,
          np.dtype(np.int32): _host.DT_INT32, np.dtype(np.int64): _host.DT_INT64}


def write_bundle(prefix: str, tensors: Dict[str, np.ndarray]) -> None:
    w = _host.BundleWriter(prefix)
    for name, arr in tensors.items():
        shape = list(np.shape(arr))                 # taken first: ascontiguousarray turns 0-d into 1-d
        arr = np.ascontiguousarray(arr)
        w.add(name, _DT_OF[arr.dtype], shape, arr)
    w.finish()


def read_bundle(prefix: str, names: Optional[List[str]] = None) -> Dict[str, np.ndarray]:
    r = _host.BundleReader(prefix)
    out = {}
    for name in (names if names is not None else r.names()):
        dtype, shape, _, _, _ = r.entry(name)
        out[name] = np.frombuffer(r.read(name), dtype=_NP_DTYPES[dtype]).reshape(shape).copy()
    return out


def bundle_names(prefix: str) -> List[str]:
    return _host.BundleReader(prefix).names()


# ---- checkpoint state ("checkpoint" file) -----------------------------------------------------
def get_checkpoint_state(checkpoint_dir: str) -> Optional[Tuple[str, List[str]]]:
    """(model_checkpoint_path, all_model_checkpoint_paths) with paths made absolute, or None."""
    st = _host.read_checkpoint_state(checkpoint_dir)
    if st is None:
        return None
    latest, allp = st

    def fix(p: str) -> str:
        return p if os.path.isabs(p) else os.path.join(checkpoint_dir, p)

    return fix(latest), [fix(p) for p in allp]


def latest_checkpoint(checkpoint_dir: str) -> Optional[str]:
    st = get_checkpoint_state(checkpoint_dir)
    if st is None:
        return None
    prefix = st[0]
    return prefix if os.path.exists(prefix + ".index") else None


def step_of(prefix: str) -> int:
    """Global step encoded in the checkpoint path suffix (reference: nn_eval.py:84)."""
    m = re.search(r"-(\d+)$", prefix)
    return int(m.group(1)) if m else -1


class Saver:
    """Saves / restores a replica's arena + optimizer state in TF bundle form."""

    def __init__(self, arena: ParamArena, optimizer: str = "adam", names: str = "tf_auto", max_to_keep: int = 5,
                 beta1: float = 0.9, beta2: float = 0.999):
        if names not in ("tf_auto", "scoped"):
            raise ValueError("names must be 'tf_auto' or 'scoped'")
        self.arena, self.optimizer, self.names = arena, optimizer, names
        self.max_to_keep = max_to_keep
        self.beta1, self.beta2 = beta1, beta2
        self._kept: List[str] = []
        self._lock = threading.Lock()

    def var_name(self, spec) -> str:
        return spec.tf_name if (self.names == "tf_auto" and spec.tf_name) else spec.name

    def name_map(self) -> Dict[str, Tuple[int, Tuple[int, ...]]]:
        """checkpoint variable name -> (flat offset, shape): the variable-name -> tensor map."""
        return {self.var_name(s): (self.arena.offsets[s.name], tuple(s.shape)) for s in self.arena.specs}

    # -- snapshot / write -------------------------------------------------------------------------
    def snapshot(self, global_step: int, updates_applied: Optional[int] = None) -> Dict[str, np.ndarray]:
        """Host copy of everything a checkpoint holds (one D2H transfer per flat buffer)."""
        a = self.arena
        host_p = a.params.detach().cpu().numpy()
        host_slots = {k: v.detach().cpu().numpy() for k, v in a.slots.items()}
        t = global_step if updates_applied is None else updates_applied
        out: Dict[str, np.ndarray] = {"global_step": np.array(global_step, dtype=np.int32)}
        for s in a.specs:
            off = a.offsets[s.name]
            name = self.var_name(s)
            out[name] = host_p[off:off + s.numel].reshape(s.shape)
            if self.optimizer == "adam" and "m" in host_slots:
                out[f"{name}/Adam"] = host_slots["m"][off:off + s.numel].reshape(s.shape)
                out[f"{name}/Adam_1"] = host_slots["v"][off:off + s.numel].reshape(s.shape)
            elif self.optimizer == "momentum" and "m" in host_slots:
                out[f"{name}/Momentum"] = host_slots["m"][off:off + s.numel].reshape(s.shape)
        if self.optimizer == "adam":
            out["beta1_power"] = np.array(self.beta1 ** (t + 1), dtype=np.float32)
            out["beta2_power"] = np.array(self.beta2 ** (t + 1), dtype=np.float32)
        return out

    def save(self, train_dir: str, global_step: int, updates_applied: Optional[int] = None,
             snapshot: Optional[Dict[str, np.ndarray]] = None) -> str:
        os.makedirs(train_dir, exist_ok=True)
        snap = snapshot if snapshot is not None else self.snapshot(global_step, updates_applied)
        prefix = os.path.join(train_dir, f"{CKPT_BASENAME}-{global_step}")
        with self._lock:
            write_bundle(prefix, snap)
            # the MetaGraphDef TensorFlow's Saver puts next to every bundle (variable list + SaverDef)
            trainable = {self.var_name(s) for s in self.arena.specs if s.trainable}
            write_meta(prefix, [(n, _DT_OF[np.asarray(t).dtype], list(np.shape(t)), n in trainable) for n, t in snap.items()],
                       self.max_to_keep)
            if prefix in self._kept:
                self._kept.remove(prefix)
            self._kept.append(prefix)
            while self.max_to_keep and len(self._kept) > self.max_to_keep:
                old = self._kept.pop(0)
                for f in glob.glob(old + ".*"):
                    try:
                        os.unlink(f)
                    except OSError:
                        pass
            rel = [os.path.basename(p) for p in self._kept]
            _host.write_checkpoint_state(train_dir, rel[-1], rel)
        return prefix

    # -- restore ------------------------------------------------------------------------------------
    def restore(self, prefix: str, strict: bool = True) -> int:
        """Loads parameters (+ optimizer slots when present); returns the stored global_step."""
        a = self.arena
        avail = set(bundle_names(prefix))
        data = read_bundle(prefix)
        host_p = a.params.detach().cpu().clone()
        host_slots = {k: v.detach().cpu().clone() for k, v in a.slots.items()}
        for s in a.specs:
            off = a.offsets[s.name]
            name = None
            for cand in (self.var_name(s), s.tf_name, s.name):
                if cand and cand in avail:
                    name = cand
                    break
            if name is None:
                if strict:
                    raise KeyError(f"checkpoint {prefix} has no tensor for {s.name} ({s.tf_name})")
                continue
            arr = data[name]
            if tuple(arr.shape) != tuple(s.shape):
                raise ValueError(f"{name}: checkpoint shape {arr.shape} != {s.shape}")
            host_p[off:off + s.numel] = torch.from_numpy(arr.reshape(-1).astype(np.float32))
            for slot, suffix in (("m", "/Adam"), ("v", "/Adam_1")) if self.optimizer == "adam" else (("m", "/Momentum"),):
                key = name + suffix
                if slot in host_slots and key in avail:
                    host_slots[slot][off:off + s.numel] = torch.from_numpy(data[key].reshape(-1).astype(np.float32))
        a.params.copy_(host_p)
        for k, v in host_slots.items():
            a.slots[k].copy_(v)
        return int(data["global_step"]) if "global_step" in data else step_of(prefix)
