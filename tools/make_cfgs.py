#!/usr/bin/env python3
"""Writes the experiment cfg files of the three workloads for one 8-GPU node.

The reference ships hand-edited cfg files for its EC2 sweeps (backup-worker counts, interval
lengths, compute-time CDF; reference: DistributedLeNet/cfg/50_workers/*, cfg/time_cdf_cfgs/*).
The same sweeps are generated here for local ranks: 1 master + 7 workers on the 8 GPUs, the swept
variable in the file name. The files are Python-literal `Cfg({...})` texts in the reference's
format, so `tools/node_launch.py --cfg FILE <verb>` and `tools/benchmark.py use_dir DIR` read
them like the reference's own files.
"""
from __future__ import annotations

import os
import pprint

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

COMMON = {
    "n_masters": 1, "n_workers": 7, "n_ps": 1, "n_evaluators": 1,
    "nfs_mount_point": "runs", "base_out_dir": "%(nfs_mount_point)s/%(name)s",
    "learning_rate_decay_factor": ".95", "num_epochs_per_decay": "1.0",
}

LENET_TRAIN = (
    "python src/mnist_distributed_train.py --batch_size=%(batch_size)s --initial_learning_rate=%(initial_learning_rate)s "
    "--learning_rate_decay_factor=%(learning_rate_decay_factor)s --num_epochs_per_decay=%(num_epochs_per_decay)s "
    "--train_dir=%(base_out_dir)s/train_dir --worker_hosts='WORKER_HOSTS' --ps_hosts='PS_HOSTS' --task_id=TASK_ID "
    "--max_steps=%(max_steps)s --num_replicas_to_aggregate=%(num_replicas_to_aggregate)s --job_name=JOB_NAME {extra}"
    "> %(base_out_dir)s/out_ROLE_ID 2>&1 &"
)
LENET_EVAL = ("python src/mnist_eval.py --eval_dir=%(base_out_dir)s/eval_dir --checkpoint_dir=%(base_out_dir)s/train_dir "
              "> %(base_out_dir)s/out_evaluator 2>&1 &")
RESNET_TRAIN = LENET_TRAIN.replace("mnist_distributed_train.py", "resnet_distributed_train.py").replace(
    "--job_name=JOB_NAME", "--num_residual_blocks=%(num_residual_blocks)s --job_name=JOB_NAME")
RESNET_EVAL = ("python src/resnet_eval.py --eval_dir=%(base_out_dir)s/eval_dir --checkpoint_dir=%(base_out_dir)s/train_dir "
               "--num_residual_blocks=%(num_residual_blocks)s > %(base_out_dir)s/out_evaluator 2>&1 &")


def write(path: str, d: dict) -> None:
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write("Cfg(" + pprint.pformat(d, width=160, sort_dicts=False) + ")\n")


def lenet(name: str, extra: str, **over) -> dict:
    d = dict(COMMON, name=name, batch_size="256", max_steps="1500", initial_learning_rate=".001",
             num_replicas_to_aggregate="8")
    d.update(over)
    d["train_commands"] = [LENET_TRAIN.format(extra=extra)]
    d["evaluate_commands"] = [LENET_EVAL]
    return d


def resnet(name: str, extra: str, n: int = 3, **over) -> dict:
    d = dict(COMMON, name=name, batch_size="256", max_steps="2000", initial_learning_rate=".001",
             num_replicas_to_aggregate="8", num_residual_blocks=str(n))
    d.update(over)
    d["train_commands"] = [RESNET_TRAIN.format(extra=extra)]
    d["evaluate_commands"] = [RESNET_EVAL]
    return d


def main() -> None:
    le = os.path.join(ROOT, "DistributedLeNet", "cfg")
    re_ = os.path.join(ROOT, "DistributedResNet", "cfg")
    te = os.path.join(ROOT, "Distributed_test", "cfg")
    # LeNet: plain sync, soft-kill, backup-worker sweep, interval sweep, compute-time CDF
    write(f"{le}/8_gpus/8_gpus_sync", lenet("lenet_8_gpus_sync", ""))
    write(f"{le}/8_gpus/8_gpus_soft_kill_6", lenet("lenet_8_gpus_soft_kill_6", "--soft_kill_method=true --should_stop_worker_num=6 ",
                                                    should_stop_worker_num="6"))
    for m in (1, 2, 4, 6, 7, 8):
        write(f"{le}/8_gpus/8_gpus_{m}_aggregate_sync",
              lenet(f"lenet_8_gpus_{m}_aggregate", "--backup_worker_method=true ", num_replicas_to_aggregate=str(m),
                    initial_learning_rate=".0008", learning_rate_decay_factor="1.0", batch_size="128"))
    for ms in (5, 10, 20, 40):
        write(f"{le}/8_gpus/8_gpus_{ms}ms_interval",
              lenet(f"lenet_8_gpus_{ms}ms_interval", f"--interval_method=true --interval_ms={ms} ", batch_size="128",
                    initial_learning_rate=".0008"))
    for b in (128, 1024):
        write(f"{le}/time_cdf_cfgs/8_gpus_batch_{b}_cdf",
              lenet(f"lenet_8_gpus_batch_{b}_cdf", "--worker_times_cdf_method=true ", batch_size=str(b), max_steps="500"))
    write(f"{le}/8_gpus/8_gpus_async_tau16", lenet("lenet_8_gpus_async_tau16", "--async_method=true --staleness_bound=16 "))
    # ResNet: the reference's default (backup workers, all aggregated), straggler drop, sharded optimizer, large batch
    write(f"{re_}/8_gpus/resnet20_sync", resnet("resnet20_8_gpus_sync", ""))
    write(f"{re_}/8_gpus/resnet32_backup_7_of_8", resnet("resnet32_8_gpus_backup_7_of_8", "--backup_worker_method=true ", n=5,
                                                          num_replicas_to_aggregate="7"))
    write(f"{re_}/8_gpus/resnet32_async_tau16", resnet("resnet32_8_gpus_async_tau16", "--async_method=true --staleness_bound=16 ", n=5))
    write(f"{re_}/8_gpus/resnet20_sharded_optimizer", resnet("resnet20_8_gpus_sharded", "--reduce_scatter=true "))
    write(f"{re_}/8_gpus/resnet20_p2p_allreduce", resnet("resnet20_8_gpus_p2p", "--allreduce_impl=p2p "))
    write(f"{re_}/8_gpus/resnet110_large_batch", resnet("resnet110_8_gpus_batch_1024", "--grad_buckets=4 ", n=18, batch_size="1024"))
    write(f"{re_}/local_8gpu.cfg", resnet("resnet20_local_8gpu", "--backup_worker_method=true "))
    # Distributed_test: 1 master + 3 workers, aggregate 4, CDF + weighted gradients
    t = lenet("test_4_gpus_cdf_weighted", "--worker_times_cdf_method=true ", n_workers=3, num_replicas_to_aggregate="4")
    write(f"{te}/4_gpus_cdf", t)
    t2 = lenet("test_4_gpus_weighted_kill_1", "--weighted_grads_method=true --num_worker_kill=1 ", n_workers=3,
               num_replicas_to_aggregate="4")
    write(f"{te}/4_gpus_weighted_kill_1", t2)


if __name__ == "__main__":
    main()
