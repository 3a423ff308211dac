#!/usr/bin/env python3
"""ResNet / CIFAR-10 training entry point, command-line compatible with the reference script of
the same name (reference: DistributedResNet/src/resnet_distributed_train.py)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))

from distributed_tensorflow_amd.flags import FLAGS  # noqa: E402
from distributed_tensorflow_amd.train.driver import train  # noqa: E402


def main(argv=None):
    FLAGS.parse(argv)
    assert FLAGS.job_name in ("ps", "worker", ""), "job_name must be ps or worker"
    return train(FLAGS, "resnet")


if __name__ == "__main__":
    main()
