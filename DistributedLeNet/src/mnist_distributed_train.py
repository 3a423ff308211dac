#!/usr/bin/env python3
"""LeNet / MNIST training entry point, command-line compatible with the reference script of the
same name (reference: DistributedLeNet/src/mnist_distributed_train.py): one process per task,
`--job_name worker|ps --task_id k --ps_hosts ... --worker_hosts ...` plus the training flags."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))

from distributed_tensorflow_amd.flags import FLAGS  # noqa: E402
from distributed_tensorflow_amd.train.driver import train  # noqa: E402


def main(argv=None):
    FLAGS.parse(argv)
    assert FLAGS.job_name in ("ps", "worker", ""), "job_name must be ps or worker"
    result = train(FLAGS, "lenet")
    return result


if __name__ == "__main__":
    main()
