"""Self-launch of one rank per GPU on the local node.

`python bench.py --gpus N` (and any other entry point that wants N local ranks without torchrun)
calls `spawn_local_ranks` from a parent process that has NOT touched the GPU (it counts the devices
in the kernel driver's sysfs tree, not through the HIP runtime): the parent only
starts N fresh children of the same command line with the torchrun environment contract
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT) and relays rank 0's standard output.
It never replaces itself with another program, so a process that has initialised the GPU is never
exec'ed over.

The reference starts its ranks with one ssh command per EC2 instance
(reference: DistributedLeNet/tools/tf_ec2.py:479-537); on one node that is a local process per GPU.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import threading
import time
from typing import Dict, List, Optional, Sequence


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launched_by_torchrun(env: Optional[Dict[str, str]] = None) -> bool:
    env = os.environ if env is None else env
    return "RANK" in env and "WORLD_SIZE" in env


KFD_NODES = "/sys/class/kfd/kfd/topology/nodes"


def _mask_count(value: str) -> int:
    """Entries of a HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES style list (the list ends at the
    first entry that is not a device: '-1' hides everything behind it)."""
    n = 0
    for item in value.split(","):
        item = item.strip()
        if not item or item.startswith("-"):
            break
        n += 1
    return n


def visible_gpus(env: Optional[Dict[str, str]] = None, kfd_nodes: str = KFD_NODES) -> int:
    """Number of GPUs the rank processes will see, found WITHOUT loading the HIP runtime: the launcher
    parent goes on to start children and must not have touched the GPU. The kernel driver lists its
    compute nodes in sysfs (a node with `simd_count` > 0 is a GPU, the others are CPUs); the
    visibility variables then cut that list down. Returns 0 when there is no amdgpu driver."""
    env = os.environ if env is None else env
    total = 0
    try:
        for node in os.listdir(kfd_nodes):
            try:
                with open(os.path.join(kfd_nodes, node, "properties")) as f:
                    for line in f:
                        key, _, val = line.partition(" ")
                        if key == "simd_count":
                            total += int(val) > 0
                            break
            except (OSError, ValueError):
                continue
    except OSError:
        return 0
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        if var in env:
            total = min(total, _mask_count(env[var]))
    return total


def rank_environment(rank: int, world: int, port: int, devices: int, base: Optional[Dict[str, str]] = None) -> Dict[str, str]:
    env = dict(os.environ if base is None else base)
    env.update({
        "RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
        "LOCAL_RANK": str(rank % devices if devices > 0 else rank),
        "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
        "DTF_SELF_LAUNCHED": "1",
    })
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return env


def spawn_local_ranks(argv: Sequence[str], world: int, *, devices: Optional[int] = None, timeout_s: float = 3000.0,
                      poll_s: float = 0.05, env: Optional[Dict[str, str]] = None) -> int:
    """Starts `world` children running `argv`, relays rank 0's stdout to ours (every rank's stderr
    goes to our stderr), and returns 0 only if every child exited 0. When one child fails the
    others are given a few seconds and then stopped (each by its own pid)."""
    if world < 1:
        raise ValueError("world must be >= 1")
    devices = visible_gpus() if devices is None else devices
    port = free_port()
    procs: List[subprocess.Popen] = []
    for r in range(world):
        procs.append(subprocess.Popen(
            list(argv), env=rank_environment(r, world, port, devices, env),
            stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=None, text=True))

    def relay() -> None:
        assert procs[0].stdout is not None
        for line in procs[0].stdout:
            sys.stdout.write(line)
            sys.stdout.flush()

    t = threading.Thread(target=relay, daemon=True)
    t.start()
    deadline = time.time() + timeout_s
    failed: Optional[int] = None
    grace_until: Optional[float] = None
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        bad = [i for i, c in enumerate(codes) if c not in (None, 0)]
        now = time.time()
        if bad and failed is None:
            failed = bad[0]
            grace_until = now + 5.0
        if (grace_until is not None and now > grace_until) or now > deadline:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()       # reaped: its exit status is known below
            break
        time.sleep(poll_s)
    t.join(timeout=5)
    codes = [p.returncode for p in procs]
    if any(c != 0 for c in codes):
        sys.stderr.write("local launch failed: exit codes by rank %s\n" % codes)
        first = codes[failed] if failed is not None else next(c for c in codes if c != 0)
        return first if isinstance(first, int) and 0 < first < 256 else 1
    return 0
