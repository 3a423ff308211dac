"""MI355X-native distributed CNN training framework (PyTorch-ROCm + hand-written HIP for gfx950
+ RCCL over xGMI) with the capabilities of hwang595/distributed_tensorflow: LeNet/MNIST and
ResNet/CIFAR-10 workloads, synchronous / backup-worker / interval / soft-kill / weighted /
asynchronous data-parallel SGD, TF-bundle checkpoints, continuous evaluation and the cfg-driven
launch surface."""
# torch must be imported before the HIP extension so that both share torch's HIP runtime.
import torch as _torch  # noqa: F401

__version__ = "0.1.0"
