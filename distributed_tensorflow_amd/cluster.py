"""Cluster description and role dispatch.

The reference starts one process per EC2 host with `--job_name {ps,worker} --task_id k
--ps_hosts ... --worker_hosts ...` and builds a tf.train.ClusterSpec / Server from them
(reference: DistributedLeNet/src/mnist_distributed_train.py:15-46). Here a worker task is one
rank of a torch.distributed job (one process per MI355X; worker hosts that name several machines
give every task its rank among the workers of its own machine); there is no parameter
server process because parameters live replicated (or sharded) in the GPUs' HBM. `--job_name=ps`
is still accepted so reference command lines run unchanged: such a process registers itself and
idles until the workers finish.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Dict, List


def _split_hosts(s: str) -> List[str]:
    return [h.strip() for h in s.split(",") if h.strip()]


@dataclass
class ClusterSpec:
    jobs: Dict[str, List[str]] = field(default_factory=dict)

    @classmethod
    def from_flags(cls, ps_hosts: str, worker_hosts: str) -> "ClusterSpec":
        return cls({"ps": _split_hosts(ps_hosts), "worker": _split_hosts(worker_hosts)})

    def as_dict(self) -> Dict[str, List[str]]:
        return {k: list(v) for k, v in self.jobs.items()}

    def num_tasks(self, job: str) -> int:
        return len(self.jobs.get(job, []))

    @property
    def num_workers(self) -> int:
        return self.num_tasks("worker")

    @property
    def num_ps(self) -> int:
        return self.num_tasks("ps")

    def task_address(self, job: str, task: int) -> str:
        return self.jobs[job][task]

    def rendezvous(self) -> tuple[str, int]:
        """Master address/port of the torch.distributed rendezvous: worker task 0's host:port."""
        host, _, port = self.jobs["worker"][0].rpartition(":")
        return (host or "127.0.0.1"), int(port)


@dataclass
class TaskEnv:
    """What a process needs to join the job, resolved from flags and/or torchrun's environment."""

    job_name: str
    task_id: int
    num_workers: int
    num_ps: int
    master_addr: str
    master_port: int
    local_rank: int = -1     # position among the workers of this task's machine (-1: not derived from host lists)
    local_world: int = -1    # workers on this task's machine

    @property
    def is_chief(self) -> bool:
        return self.job_name == "worker" and self.task_id == 0

    def export(self) -> None:
        """Exports the torchrun contract so that Communicator.from_env() picks it up."""
        os.environ["RANK"] = str(self.task_id)
        os.environ["WORLD_SIZE"] = str(self.num_workers)
        os.environ.setdefault("LOCAL_RANK", str(self.local_rank if self.local_rank >= 0 else self.task_id))
        if self.local_world > 0:
            # worker hosts that name several machines: the driver then coordinates through the process
            # group's store instead of shared memory (LOCAL_WORLD_SIZE < WORLD_SIZE)
            os.environ.setdefault("LOCAL_WORLD_SIZE", str(self.local_world))
        os.environ["MASTER_ADDR"] = self.master_addr
        os.environ["MASTER_PORT"] = str(self.master_port)


def resolve_task(job_name: str, task_id: int, ps_hosts: str, worker_hosts: str) -> TaskEnv:
    """Flags win; without host lists, fall back to torchrun's RANK/WORLD_SIZE (or one replica)."""
    spec = ClusterSpec.from_flags(ps_hosts, worker_hosts)
    if spec.num_workers > 0:
        addr, port = spec.rendezvous()
        if addr in ("localhost", ""):
            addr = "127.0.0.1"
        # Twisted RPC and the TF server shared the worker port in the reference; the torch store
        # gets its own port derived from it so both conventions can coexist on one machine.
        hosts = [h.rpartition(":")[0] or "127.0.0.1" for h in spec.jobs["worker"]]
        hosts = ["127.0.0.1" if h == "localhost" else h for h in hosts]
        local_rank = local_world = -1
        if (job_name or "worker") == "worker" and 0 <= task_id < len(hosts):
            mine = hosts[task_id]
            local_world = sum(1 for h in hosts if h == mine)
            local_rank = sum(1 for h in hosts[:task_id] if h == mine)
        return TaskEnv(job_name or "worker", task_id, spec.num_workers, spec.num_ps, addr, port + 1000 if port < 64000 else port - 1000,
                       local_rank, local_world)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", str(task_id)))
    return TaskEnv(job_name or "worker", rank, world, 0, os.environ.get("MASTER_ADDR", "127.0.0.1"),
                   int(os.environ.get("MASTER_PORT", "29500")))
