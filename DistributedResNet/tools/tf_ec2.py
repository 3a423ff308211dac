#!/usr/bin/env python3
"""`python tools/tf_ec2.py <command>` for this workload: the reference's cluster tool, acting on
the local GPUs (see tools/node_launch.py at the repository root for the verbs). Without --cfg the
workload's default cfg in ../cfg is used."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import node_launch  # noqa: E402

DEFAULTS = {"DistributedLeNet": "cfg/8_gpus/8_gpus_soft_kill_6", "DistributedResNet": "cfg/local_8gpu.cfg",
            "Distributed_test": "cfg/4_gpus_cdf"}

if __name__ == "__main__":
    if "--cfg" not in sys.argv:
        workload = os.path.basename(os.path.dirname(HERE))
        sys.argv[1:1] = ["--cfg", os.path.join(os.path.dirname(HERE), DEFAULTS[workload])]
    node_launch.main()
