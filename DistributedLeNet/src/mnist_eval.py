#!/usr/bin/env python3
"""LeNet / MNIST continuous evaluator (reference: DistributedLeNet/src/mnist_eval.py)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))

from distributed_tensorflow_amd.flags import FLAGS  # noqa: E402
from distributed_tensorflow_amd.train.evaluator import evaluate  # noqa: E402


def main(argv=None):
    FLAGS.parse(argv)
    return evaluate(FLAGS, "lenet")


if __name__ == "__main__":
    main()
