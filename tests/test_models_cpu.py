"""Model definitions, LR schedule, data pipelines and reference-op semantics on CPU."""
import gzip
import os
import pickle
import struct

import numpy as np
import pytest
import torch

from distributed_tensorflow_amd import lr as lr_mod
from distributed_tensorflow_amd.data import cifar10 as cifar
from distributed_tensorflow_amd.data import mnist
from distributed_tensorflow_amd.models import lenet, resnet
from distributed_tensorflow_amd.ops import reference as ref
from distributed_tensorflow_amd.runtime.arena import ALIGN_FLOATS, ParamArena


def test_parameter_counts_match_the_reference_models():
    assert sum(s.numel for s in lenet.lenet_param_specs()) == 1663370
    assert sum(s.numel for s in resnet.resnet_param_specs(3)) == 269722
    assert sum(s.numel for s in resnet.resnet_param_specs(18)) == 1727962
    assert len(resnet.resnet_param_specs(18)) == 329
    assert resnet.num_layers(18) == 110


def test_resnet_creation_order_follows_the_reference():
    names = [s.name for s in resnet.resnet_param_specs(2)]
    # conv_bn_relu: filter, beta, gamma; first block conv1 has no BN; bn_relu_conv: beta, gamma, filter
    assert names[:9] == [
        "conv0/conv", "conv0/beta", "conv0/gamma",
        "conv1_0/conv1_in_block/conv",
        "conv1_0/conv2_in_block/beta", "conv1_0/conv2_in_block/gamma", "conv1_0/conv2_in_block/conv",
        "conv1_1/conv1_in_block/beta", "conv1_1/conv1_in_block/gamma",
    ]
    assert names[-4:] == ["fc/beta", "fc/gamma", "fc/fc_weights", "fc/fc_bias"]
    blocks = resnet.resnet_blocks(2)
    assert [b.conv1.stride for b in blocks] == [1, 1, 2, 1, 2, 1]
    assert [b.increase_dim for b in blocks] == [False, False, True, False, True, False]
    assert blocks[0].first_block and blocks[0].conv1.bn_beta is None


def test_forward_shapes_and_init():
    a = ParamArena(resnet.resnet_param_specs(1), "cpu")
    a.initialize()
    assert a.total % ALIGN_FLOATS == 0 and all(o % ALIGN_FLOATS == 0 for o in a.offsets.values())
    w = a.param("conv0/conv")
    assert float(w.abs().max()) <= 0.2 + 1e-6 and 0.05 < float(w.std()) < 0.12    # truncated normal, sigma 0.1
    assert torch.all(a.param("conv0/gamma") == 1) and torch.all(a.param("conv0/beta") == 0)
    x = torch.randn(4, 32, 32, 3)
    assert resnet.resnet_forward(a.param, x, 1).shape == (4, 10)
    la = ParamArena(lenet.lenet_param_specs(), "cpu")
    la.initialize()
    assert torch.all(la.param("conv1_biases") == 0) and torch.all(la.param("fc2_biases") == 0.1)
    xm = torch.rand(3, 28, 28, 1) - 0.5
    assert lenet.lenet_forward(la.param, xm).shape == (3, 10)
    # dropout only in training, scaled by 1/keep
    g = torch.Generator().manual_seed(1)
    l_train = lenet.lenet_forward(la.param, xm, train=True, dropout_gen=g)
    assert not torch.allclose(l_train, lenet.lenet_forward(la.param, xm))


def test_tf_same_padding_semantics():
    assert ref.same_pad(32, 3, 1) == (32, 1, 1)
    assert ref.same_pad(32, 3, 2) == (16, 0, 1)      # extra padding goes at the bottom/right
    assert ref.same_pad(28, 5, 1) == (28, 2, 2)
    x = torch.arange(16, dtype=torch.float32).reshape(1, 4, 4, 1)
    w = torch.ones(3, 3, 1, 1)
    y = ref.conv2d_same(x, w, 2)
    assert y.shape == (1, 2, 2, 1)
    assert float(y[0, 0, 0, 0]) == float(x[0, 0:3, 0:3, 0].sum())      # window starts at 0, not -1
    assert float(y[0, 1, 1, 0]) == float(x[0, 2:4, 2:4, 0].sum())
    p = ref.max_pool_2x2_same(torch.arange(9, dtype=torch.float32).reshape(1, 3, 3, 1))
    assert p.reshape(-1).tolist() == [4, 5, 7, 8]
    s = ref.shortcut_a(torch.ones(1, 4, 4, 8))
    assert s.shape == (1, 2, 2, 16) and float(s[..., :4].abs().sum()) == 0 and float(s[..., 4:12].min()) == 1.0


def test_batchnorm_uses_batch_statistics():
    x = torch.randn(8, 4, 4, 6) * 3 + 2
    y = ref.batch_norm(x, torch.ones(6), torch.zeros(6))
    flat = y.reshape(-1, 6)
    assert torch.allclose(flat.mean(0), torch.zeros(6), atol=1e-5)
    var = x.reshape(-1, 6).var(0, unbiased=False)
    assert torch.allclose(flat.var(0, unbiased=False), var / (var + ref.BN_EPSILON), atol=1e-4)


def test_lr_schedule():
    # decay period divided by the number of aggregated replicas (reference: distributed_train.py:143)
    assert lr_mod.decay_steps(60000, 128, 2.0, 8) == int(60000 / 128 * 2.0 / 8)
    d = lr_mod.decay_steps(60000, 256, 1.0, 8)
    assert lr_mod.exponential_decay(1e-3, d - 1, d, 0.95) == 1e-3
    assert abs(lr_mod.exponential_decay(1e-3, d, d, 0.95) - 0.95e-3) < 1e-12
    assert abs(lr_mod.exponential_decay(1e-3, 3 * d + 1, d, 0.95) - 1e-3 * 0.95 ** 3) < 1e-12
    assert abs(ref.adam_lr_t(1e-3, 1) - 1e-3 * (1 - 0.999) ** 0.5 / 0.1) < 1e-12


def _write_idx(path, magic, arr):
    with gzip.open(path, "wb") as f:
        f.write(struct.pack(">I", magic))
        for d in arr.shape:
            f.write(struct.pack(">I", d))
        f.write(arr.astype(np.uint8).tobytes())


def test_mnist_pipeline_from_idx_files(tmp_path):
    rng = np.random.RandomState(0)
    tr_x, tr_y = rng.randint(0, 256, (300, 28, 28)), rng.randint(0, 10, 300)
    te_x, te_y = rng.randint(0, 256, (100, 28, 28)), rng.randint(0, 10, 100)
    _write_idx(tmp_path / mnist.TRAIN_IMAGES, 2051, tr_x)
    _write_idx(tmp_path / mnist.TRAIN_LABELS, 2049, tr_y)
    _write_idx(tmp_path / mnist.TEST_IMAGES, 2051, te_x)
    _write_idx(tmp_path / mnist.TEST_LABELS, 2049, te_y)
    sets = mnist.load_mnist(str(tmp_path))
    assert sets.train.num_examples == 300 and sets.validation.num_examples == 100   # validation = test set
    x, y = sets.train.next_batch(64)
    assert x.shape == (64, 28, 28, 1) and x.dtype == np.float32 and y.dtype == np.int64
    assert x.min() >= -0.5 and x.max() <= 0.5
    # two workers iterate the same sequence (fixed seed, no sharding)
    again = mnist.load_mnist(str(tmp_path), worker_id=1, n_workers=2)
    x2, y2 = again.train.next_batch(64)
    assert np.array_equal(x, x2) and np.array_equal(y, y2)
    for _ in range(5):
        sets.train.next_batch(64)
    assert sets.train.epochs_completed == 1
    fake = mnist.load_mnist("")
    fx, fy = fake.train.next_batch(8)
    assert fx.shape == (8, 28, 28, 1) and fy.shape == (8,)


def test_cifar_pipeline(tmp_path):
    rng = np.random.RandomState(0)
    for name in cifar.TRAIN_FILES + [cifar.TEST_FILE]:
        with open(tmp_path / name, "wb") as f:
            pickle.dump({b"data": rng.randint(0, 256, (20, 3072)).astype(np.uint8), b"labels": [int(v) for v in rng.randint(0, 10, 20)]}, f, protocol=2)
    data, labels = cifar.prepare_train_data(str(tmp_path), 2)
    assert data.shape == (100, 36, 36, 3) and labels.shape == (100,)
    assert float(np.abs(data[:, :2]).sum()) == 0       # zero padding
    b = cifar.TrainBatcher(data, labels, 2, seed=3)
    x, y = b.next_batch(16)
    assert x.shape == (16, 32, 32, 3) and y.shape == (16,)
    assert np.allclose(x.reshape(16, -1).mean(1), 0, atol=1e-4) and np.allclose(x.reshape(16, -1).std(1), 1, atol=1e-3)
    vx, vy = cifar.read_validation_data(str(tmp_path))
    assert vx.shape == (20, 32, 32, 3)
    t = ref.per_image_whiten(torch.from_numpy(rng.rand(4, 32, 32, 3).astype(np.float32) * 255))
    assert torch.allclose(t.reshape(4, -1).std(1, unbiased=False), torch.ones(4), atol=1e-3)
    # a pickle that references anything but arrays/containers is refused
    class Evil:
        def __reduce__(self):
            return (os.system, ("true",))
    with open(tmp_path / "evil", "wb") as f:
        pickle.dump({b"data": Evil()}, f)
    with pytest.raises(pickle.UnpicklingError):
        cifar._read_one_batch(str(tmp_path / "evil"))
    sx, sy = cifar.prepare_train_data("", 2, num_synthetic=64)
    assert sx.shape == (64, 36, 36, 3)


def test_mnist_device_batcher_follows_the_host_dataset():
    """The device-resident batcher returns exactly the batches DataSet.next_batch returns,
    across epoch wraps (same seed-448 reshuffle of the current order)."""
    import numpy as np
    import torch

    from distributed_tensorflow_amd.data.mnist import DataSet, GpuBatcher

    rng = np.random.RandomState(0)
    images = rng.rand(50, 28, 28, 1).astype(np.float32)
    labels = rng.randint(0, 10, 50).astype(np.int64)
    host = DataSet(images.copy(), labels.copy())
    dev = GpuBatcher(DataSet(images.copy(), labels.copy()), "cpu")
    xb = torch.empty(16, 28, 28, 1)
    yb = torch.empty(16, dtype=torch.int32)
    for _ in range(11):   # 3.5 epochs
        hx, hy = host.next_batch(16)
        dev.next_batch_into(xb, yb)
        assert np.array_equal(xb.numpy(), hx) and np.array_equal(yb.numpy(), hy.astype(np.int32))
    assert dev.epochs_completed == host.epochs_completed == 3
    # fake-data mode: same stream on every worker, fresh batch per step
    f1, f2 = GpuBatcher(DataSet(None, None, True, 100), "cpu"), GpuBatcher(DataSet(None, None, True, 100), "cpu")
    x1, x2, y1, y2 = torch.empty(8, 28, 28, 1), torch.empty(8, 28, 28, 1), torch.empty(8, dtype=torch.int32), torch.empty(8, dtype=torch.int32)
    f1.next_batch_into(x1, y1)
    f2.next_batch_into(x2, y2)
    assert torch.equal(x1, x2) and torch.equal(y1, y2) and float(x1.min()) >= -0.5 and float(x1.max()) <= 0.5
    prev = x1.clone()
    f1.next_batch_into(x1, y1)
    assert not torch.equal(prev, x1)


def test_histogram_summary_round_trip(tmp_path):
    """Histogram events: TensorFlow's default buckets, empty runs collapsed, readable back."""
    import numpy as np

    from distributed_tensorflow_amd.utils import summary as sm

    limits = sm.default_bucket_limits()
    assert limits[0] < -1e300 and limits[-1] > 1e300 and 0.0 in limits and np.all(np.diff(limits) > 0)
    v = np.array([0.0, 0.0, 1.0, 1.05, -3.0, 5e-13], dtype=np.float32)
    h = sm.histogram_of(v)
    assert h["num"] == 6 and h["min"] == -3.0 and sum(h["bucket"]) == 6
    # a value lands in the first bucket whose limit exceeds it: zeros go to the bucket ending at 1e-12
    i = h["bucket_limit"].index(1.0e-12)
    assert h["bucket"][i] == 3          # the two zeros and 5e-13
    assert len(h["bucket"]) < 16        # empty runs are collapsed
    w = sm.FileWriter(str(tmp_path))
    w.add_scalars({"loss": 0.25}, 3)
    w.add_histograms({"conv0/activations": h}, 3)
    w.close()
    assert list(sm.read_events(w.path)) == [(3, {"loss": 0.25})]
    assert list(sm.read_histograms(w.path)) == [(3, {"conv0/activations": h})]
