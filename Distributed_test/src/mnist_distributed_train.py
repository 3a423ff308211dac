#!/usr/bin/env python3
"""Distributed_test workload: LeNet / MNIST with the experimental aggregation policies
(worker-time CDF tracking and least-squares weighted gradients under simulated worker kills;
reference: Distributed_test/src/mnist_distributed_train.py, Distributed_test/src/distributed_train.py).
Unless a method flag says otherwise this entry point turns the weighted-gradient experiment on,
which is what distinguishes the reference's Distributed_test driver from the LeNet one."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))

from distributed_tensorflow_amd.flags import FLAGS  # noqa: E402
from distributed_tensorflow_amd.train.driver import train  # noqa: E402


def main(argv=None):
    FLAGS.parse(argv)
    explicit = any(FLAGS.is_present(n) for n in ("interval_method", "worker_times_cdf_method", "backup_worker_method",
                                                 "soft_kill_method", "weighted_grads_method", "async_method"))
    if not explicit:
        FLAGS.weighted_grads_method = True
    return train(FLAGS, "lenet")


if __name__ == "__main__":
    main()
