"""CIFAR-10 input pipeline (reference: DistributedResNet/src/cifar10_input.py and the batch
helpers in DistributedResNet/src/distributed_train.py:144-179).

Behaviour kept from the reference:
  * python-version batch files (`data_batch_1..5`, `test_batch`) -> NHWC float32 [N,32,32,3] via a
    Fortran-order reshape of the [N,3072] rows (cifar10_input.py:98-99), then a shuffle (:102-106);
  * training images are zero-padded by `padding_size` on each side (:146-160);
  * a training batch is a sequential slice (epoch wrap + reshuffle), a random crop back to 32x32 with
    horizontal flipping disabled (:125-143), and per-image whitening (:111-122) - in the reference the
    whitening happens through an in-place call inside a log statement
    (distributed_train.py:177); here it is an explicit step with the same result;
  * validation data is the whitened 10k test batch (:163-172).
The batch files are unpickled ONLY with a restricted unpickler that admits numpy arrays and
builtin containers, because pickle can execute code. With no files present, synthetic data of
the same shape is used. On the GPU, crop + whitening run in one HIP kernel (ops.native.crop_whiten)
over the padded images kept resident in HBM.
"""
from __future__ import annotations

import io
import os
import pickle
from typing import Optional, Tuple

import numpy as np

IMG_SIZE = 32
IMG_DEPTH = 3
NUM_CLASS = 10
TRAIN_FILES = [f"data_batch_{i}" for i in range(1, 6)]
TEST_FILE = "test_batch"


class _RestrictedUnpickler(pickle.Unpickler):
    _ALLOWED = {
        ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
        ("numpy", "ndarray"), ("numpy", "dtype"), ("builtins", "dict"), ("builtins", "list"),
        ("__builtin__", "dict"), ("__builtin__", "list"), ("builtins", "object"),
        ("_codecs", "encode"),   # how Python 3 pickles the byte payload of an array at protocol <= 2
    }

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            import importlib

            mod = "builtins" if module == "__builtin__" else module
            mod = mod.replace("numpy.core", "numpy._core") if module.startswith("numpy.core") and not _has_module(module) else mod
            return getattr(importlib.import_module(mod), name)
        raise pickle.UnpicklingError(f"refusing to unpickle {module}.{name}")


def _has_module(name: str) -> bool:
    import importlib.util

    try:
        return importlib.util.find_spec(name) is not None
    except (ImportError, ValueError):
        return False


def _read_one_batch(path: str, is_random_label: bool = False) -> Tuple[np.ndarray, np.ndarray]:
    with open(path, "rb") as f:
        d = _RestrictedUnpickler(io.BytesIO(f.read()), encoding="bytes").load()
    data = d.get(b"data", d.get("data"))
    labels = d.get(b"labels", d.get("labels"))
    if is_random_label:
        labels = np.random.randint(low=0, high=10, size=10000)
    return np.asarray(data), np.asarray(labels)


def read_in_all_images(paths, shuffle: bool = True, is_random_label: bool = False, rng: Optional[np.random.RandomState] = None):
    data = np.zeros((0, IMG_SIZE * IMG_SIZE * IMG_DEPTH), dtype=np.uint8)
    label = np.zeros((0,), dtype=np.int64)
    for p in paths:
        d, l = _read_one_batch(p, is_random_label)
        data = np.concatenate((data, d))
        label = np.concatenate((label, l))
    n = len(label)
    data = data.reshape((n, IMG_SIZE * IMG_SIZE, IMG_DEPTH), order="F").reshape((n, IMG_SIZE, IMG_SIZE, IMG_DEPTH))
    if shuffle:
        order = (rng or np.random).permutation(n)
        data, label = data[order], label[order]
    return data.astype(np.float32), label.astype(np.int32)


def whitening_image(image_np: np.ndarray) -> np.ndarray:
    """Per-image standardisation, in place like the reference, returned for convenience."""
    for i in range(len(image_np)):
        mean = np.mean(image_np[i, ...])
        std = np.max([np.std(image_np[i, ...]), 1.0 / np.sqrt(IMG_SIZE * IMG_SIZE * IMG_DEPTH)])
        image_np[i, ...] = (image_np[i, ...] - mean) / std
    return image_np


def random_crop_and_flip(batch_data: np.ndarray, padding_size: int, rng: Optional[np.random.RandomState] = None) -> np.ndarray:
    """Random 32x32 crop out of each padded image. Offsets are drawn from 0 .. 2*padding_size - 1 like
    the reference (np.random.randint(low=0, high=2*padding_size), cifar10_input.py:136-137): the
    last possible position (offset 2*padding_size, the crop flush with the bottom/right edge) is
    never used there, and is not used here either, so both pipelines sample the same set of crops."""
    rng = rng or np.random
    out = np.zeros((len(batch_data), IMG_SIZE, IMG_SIZE, IMG_DEPTH), dtype=np.float32)
    for i in range(len(batch_data)):
        x_off = rng.randint(low=0, high=max(1, 2 * padding_size))
        y_off = rng.randint(low=0, high=max(1, 2 * padding_size))
        out[i] = batch_data[i][x_off:x_off + IMG_SIZE, y_off:y_off + IMG_SIZE, :]
        # horizontal flipping is disabled in the reference (cifar10_input.py:141)
    return out


def pad_images(data: np.ndarray, padding_size: int) -> np.ndarray:
    return np.pad(data, ((0, 0), (padding_size, padding_size), (padding_size, padding_size), (0, 0)), mode="constant")


def have_files(data_dir: str) -> bool:
    return bool(data_dir) and all(os.path.exists(os.path.join(data_dir, f)) for f in TRAIN_FILES + [TEST_FILE])


def synthetic_cifar(n: int, seed: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    rng = np.random.RandomState(seed)
    return rng.uniform(0, 255, (n, IMG_SIZE, IMG_SIZE, IMG_DEPTH)).astype(np.float32), rng.randint(0, NUM_CLASS, n).astype(np.int32)


def prepare_train_data(data_dir: str, padding_size: int, num_synthetic: int = 2048) -> Tuple[np.ndarray, np.ndarray]:
    if have_files(data_dir):
        data, label = read_in_all_images([os.path.join(data_dir, f) for f in TRAIN_FILES])
    else:
        data, label = synthetic_cifar(num_synthetic, 1)
    return pad_images(data, padding_size), label


def read_validation_data(data_dir: str, num_synthetic: int = 1000) -> Tuple[np.ndarray, np.ndarray]:
    if have_files(data_dir):
        data, label = read_in_all_images([os.path.join(data_dir, TEST_FILE)], shuffle=False)
    else:
        data, label = synthetic_cifar(num_synthetic, 2)
    return whitening_image(data), label


class TrainBatcher:
    """Sequential slicing with epoch wrap and reshuffle + crop + whitening
    (reference: DistributedResNet/src/distributed_train.py:144-179)."""

    def __init__(self, data: np.ndarray, labels: np.ndarray, padding_size: int, seed: Optional[int] = None):
        self.data, self.labels, self.padding = data, labels, padding_size
        self.rng = np.random.RandomState(seed)
        self.offset = 0
        self.epochs = 0

    @property
    def num_examples(self) -> int:
        return len(self.labels)

    def next_indices(self, batch_size: int) -> np.ndarray:
        n = self.num_examples
        if self.offset + batch_size > n:
            self.epochs += 1
            self.offset = 0
            order = self.rng.permutation(n)
            self.data, self.labels = self.data[order], self.labels[order]
        idx = np.arange(self.offset, self.offset + batch_size) % n
        self.offset += batch_size
        return idx

    def next_batch(self, batch_size: int) -> Tuple[np.ndarray, np.ndarray]:
        idx = self.next_indices(batch_size)
        batch = random_crop_and_flip(self.data[idx], self.padding, self.rng)
        return whitening_image(batch), self.labels[idx]


class GpuTrainBatcher:
    """TrainBatcher with the dataset resident in HBM: the zero-padded images are uploaded once
    (50000 x 36 x 36 x 3 fp32 = 0.78 GB of the 288 GB), a step draws the same indices and crop
    offsets as TrainBatcher from the same seed, and one kernel crops + whitens straight into the
    step program's input buffer (reference: DistributedResNet/src/cifar10_input.py:111-143 on the
    host). Shuffling permutes an index array instead of the data."""

    def __init__(self, data: np.ndarray, labels: np.ndarray, padding_size: int, seed: Optional[int] = None, device="cuda"):
        import torch

        self.padding = padding_size
        self.rng = np.random.RandomState(seed)
        self.order = np.arange(len(labels))
        self.offset = 0
        self.epochs = 0
        self.src = torch.from_numpy(np.ascontiguousarray(data, dtype=np.float32)).to(device)
        self.labels = torch.from_numpy(np.asarray(labels, dtype=np.int32)).to(device)
        self._host = None

    @property
    def num_examples(self) -> int:
        return len(self.order)

    def next_indices(self, batch_size: int) -> np.ndarray:
        n = self.num_examples
        if self.offset + batch_size > n:
            self.epochs += 1
            self.offset = 0
            self.order = self.order[self.rng.permutation(n)]
        pos = np.arange(self.offset, self.offset + batch_size) % n
        self.offset += batch_size
        return self.order[pos]

    def next_batch_into(self, images_out, labels_out) -> None:
        """Fills images_out [B,32,32,3] fp32 and labels_out [B] int32 on the current stream."""
        import torch

        from ..ops import native as nv

        b = images_out.shape[0]
        idx = self.next_indices(b)
        if self._host is None or self._host.shape[0] != b:
            self._host = torch.zeros((b, 3), dtype=torch.int32).pin_memory()
        h = self._host.numpy()
        h[:, 0] = idx
        # same draw order as random_crop_and_flip, image by image: row offset, then column offset (one
        # array draw yields the same stream as 2 b scalar draws, without 2 b interpreter round trips)
        h[:, 1:3] = self.rng.randint(low=0, high=max(1, 2 * self.padding), size=(b, 2))
        dev = self._host.to(images_out.device, non_blocking=True)
        index = dev[:, 0].contiguous()
        offs = dev[:, 1:3].contiguous()
        nv.crop_whiten(self.src, index, offs, out_hw=(IMG_SIZE, IMG_SIZE), whiten=True, out=images_out)
        labels_out.copy_(self.labels[index.long()])
