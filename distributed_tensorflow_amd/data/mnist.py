"""MNIST input pipeline (reference: DistributedLeNet/src/mnist_data.py).

Behaviour kept from the reference:
  * pixels are rescaled to (x - 127.5) / 255 in [-0.5, 0.5], NHWC [N,28,28,1] (mnist_data.py:138-151);
  * labels are int64 class ids (:153-160);
  * the training set is the full 60k images and "validation" is the 10k TEST set (:206-207);
  * DataSet shuffles with the fixed numpy seed 448 at construction and at every epoch wrap
    (:60,86-89,120-128), and `worker_id` / `n_workers` are accepted but ignored (:162-169), so
    every worker walks the same batch sequence;
  * a `fake_data` mode that needs no files (:65-67,109-117).
There is no network here: files are read from `data_dir` (the four idx .gz files, or the same
files uncompressed); when they are absent the caller falls back to synthetic data.
"""
from __future__ import annotations

import gzip
import os
import struct
from collections import namedtuple
from typing import Optional, Tuple

import numpy as np

SHUFFLE_SEED = 448
TRAIN_IMAGES = "train-images-idx3-ubyte.gz"
TRAIN_LABELS = "train-labels-idx1-ubyte.gz"
TEST_IMAGES = "t10k-images-idx3-ubyte.gz"
TEST_LABELS = "t10k-labels-idx1-ubyte.gz"
PIXEL_DEPTH = 255

Datasets = namedtuple("Datasets", ["train", "validation"])


def _open(path: str):
    if os.path.exists(path):
        return gzip.open(path, "rb") if path.endswith(".gz") else open(path, "rb")
    alt = path[:-3] if path.endswith(".gz") else path + ".gz"
    if os.path.exists(alt):
        return gzip.open(alt, "rb") if alt.endswith(".gz") else open(alt, "rb")
    raise FileNotFoundError(path)


def extract_data(path: str) -> np.ndarray:
    with _open(path) as f:
        magic, n, rows, cols = struct.unpack(">IIII", f.read(16))
        if magic != 2051:
            raise ValueError(f"bad MNIST image magic {magic} in {path}")
        data = np.frombuffer(f.read(n * rows * cols), dtype=np.uint8).astype(np.float32)
    data = (data - (PIXEL_DEPTH / 2.0)) / PIXEL_DEPTH
    return data.reshape(n, rows, cols, 1)


def extract_labels(path: str) -> np.ndarray:
    with _open(path) as f:
        magic, n = struct.unpack(">II", f.read(8))
        if magic != 2049:
            raise ValueError(f"bad MNIST label magic {magic} in {path}")
        return np.frombuffer(f.read(n), dtype=np.uint8).astype(np.int64)


class DataSet:
    def __init__(self, images: Optional[np.ndarray], labels: Optional[np.ndarray], fake_data: bool = False,
                 num_fake: int = 10000):
        self.fake_data = fake_data
        if fake_data:
            self._num_examples = num_fake
            self._images = None
            self._labels = None
        else:
            assert images.shape[0] == labels.shape[0], f"images.shape: {images.shape} labels.shape: {labels.shape}"
            self._num_examples = images.shape[0]
            np.random.seed(SHUFFLE_SEED)
            perm = np.arange(self._num_examples)
            np.random.shuffle(perm)
            self._images = images[perm]
            self._labels = labels[perm]
        self._epochs_completed = 0
        self._index_in_epoch = 0

    @property
    def images(self):
        return self._images

    @property
    def labels(self):
        return self._labels

    @property
    def num_examples(self) -> int:
        return self._num_examples

    @property
    def epochs_completed(self) -> int:
        return self._epochs_completed

    def next_batch(self, batch_size: int) -> Tuple[np.ndarray, np.ndarray]:
        if self.fake_data:
            rng = np.random.RandomState(self._index_in_epoch + 7919 * self._epochs_completed)
            self._index_in_epoch += batch_size
            if self._index_in_epoch > self._num_examples:
                self._epochs_completed += 1
                self._index_in_epoch = batch_size
            return (rng.uniform(-0.5, 0.5, (batch_size, 28, 28, 1)).astype(np.float32),
                    rng.randint(0, 10, batch_size).astype(np.int64))
        start = self._index_in_epoch
        self._index_in_epoch += batch_size
        if self._index_in_epoch > self._num_examples:
            self._epochs_completed += 1
            np.random.seed(SHUFFLE_SEED)
            perm = np.arange(self._num_examples)
            np.random.shuffle(perm)
            self._images = self._images[perm]
            self._labels = self._labels[perm]
            start = 0
            self._index_in_epoch = batch_size
            assert batch_size <= self._num_examples
        end = self._index_in_epoch
        return self._images[start:end], self._labels[start:end]


class GpuBatcher:
    """The training set resident on the device: `next_batch_into` fills the step program's input
    buffers with the batch `DataSet.next_batch` would return next, without touching the host.

    The data set keeps its (already shuffled) order on the device; a batch is a contiguous slice
    copied device-to-device, and the epoch-wrap reshuffle (the same seed-448 permutation every
    time, applied to the current order; reference: DistributedLeNet/src/mnist_data.py:120-128) is
    one gather on the device. In fake-data mode batches are drawn on the device from a generator
    seeded by the position in the stream, so every worker still sees the same sequence."""

    def __init__(self, dataset: DataSet, device):
        import torch

        self._torch = torch
        self.device = torch.device(device)
        self.fake = dataset.fake_data
        self.num_examples = dataset.num_examples
        self.index_in_epoch = dataset._index_in_epoch
        self.epochs_completed = dataset.epochs_completed
        if self.fake:
            self._gen = torch.Generator(device=self.device)
        else:
            self.images = torch.as_tensor(np.ascontiguousarray(dataset.images), dtype=torch.float32).to(self.device)
            self.labels = torch.as_tensor(np.ascontiguousarray(dataset.labels)).to(self.device)
            np.random.seed(SHUFFLE_SEED)
            perm = np.arange(self.num_examples)
            np.random.shuffle(perm)
            self._perm = torch.as_tensor(perm).to(self.device)

    def next_batch_into(self, images_out, labels_out) -> None:
        torch = self._torch
        b = images_out.shape[0]
        if self.fake:
            self._gen.manual_seed(self.index_in_epoch + 7919 * self.epochs_completed)
            self.index_in_epoch += b
            if self.index_in_epoch > self.num_examples:
                self.epochs_completed += 1
                self.index_in_epoch = b
            images_out.uniform_(-0.5, 0.5, generator=self._gen)
            labels_out.copy_(torch.randint(0, 10, (b,), device=self.device, generator=self._gen))
            return
        start = self.index_in_epoch
        self.index_in_epoch += b
        if self.index_in_epoch > self.num_examples:
            assert b <= self.num_examples
            self.epochs_completed += 1
            self.images = self.images.index_select(0, self._perm)
            self.labels = self.labels.index_select(0, self._perm)
            start = 0
            self.index_in_epoch = b
        end = self.index_in_epoch
        images_out.copy_(self.images[start:end].reshape(images_out.shape))
        labels_out.copy_(self.labels[start:end])


def read_data_sets(train_dir: str, fake_data: bool = False, worker_id: int = -1, n_workers: int = -1) -> Datasets:
    """`worker_id` / `n_workers` are accepted for signature parity and, like the reference, unused."""
    if fake_data:
        return Datasets(train=DataSet(None, None, True, 60000), validation=DataSet(None, None, True, 10000))
    train_images = extract_data(os.path.join(train_dir, TRAIN_IMAGES))
    train_labels = extract_labels(os.path.join(train_dir, TRAIN_LABELS))
    test_images = extract_data(os.path.join(train_dir, TEST_IMAGES))
    test_labels = extract_labels(os.path.join(train_dir, TEST_LABELS))
    return Datasets(train=DataSet(train_images, train_labels), validation=DataSet(test_images, test_labels))


def have_files(train_dir: str) -> bool:
    if not train_dir:
        return False
    for n in (TRAIN_IMAGES, TRAIN_LABELS, TEST_IMAGES, TEST_LABELS):
        if not (os.path.exists(os.path.join(train_dir, n)) or os.path.exists(os.path.join(train_dir, n[:-3]))):
            return False
    return True


def load_mnist(data_dir: str = "", worker_id: int = -1, n_workers: int = -1) -> Datasets:
    """MNIST from files when present, else the fake-data sets of the same shape."""
    if have_files(data_dir):
        return read_data_sets(data_dir, False, worker_id, n_workers)
    return read_data_sets(data_dir, True, worker_id, n_workers)
