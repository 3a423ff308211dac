"""Helpers to run a function in N local processes joined into a gloo process group."""
import os
import socket
import traceback

import torch.multiprocessing as mp


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _entry(fn, rank, world, port, q, args):
    os.environ["RANK"] = str(rank)
    os.environ["LOCAL_RANK"] = str(rank)
    os.environ["WORLD_SIZE"] = str(world)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "1"
    try:
        import torch

        torch.set_num_threads(1)
        q.put((rank, "ok", fn(rank, world, *args)))
    except BaseException as e:  # noqa: BLE001
        q.put((rank, "err", "".join(traceback.format_exception(type(e), e, e.__traceback__))))


def run_workers(fn, world, *args, timeout=240):
    """Runs fn(rank, world, *args) in `world` spawned processes; returns the results by rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_entry, args=(fn, r, world, port, q, args)) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    try:
        for _ in range(world):
            rank, status, payload = q.get(timeout=timeout)
            if status != "ok":
                raise AssertionError(f"rank {rank} failed:\n{payload}")
            results[rank] = payload
    finally:
        for p in procs:
            p.join(timeout=20)
            if p.is_alive():
                p.terminate()
    return [results[r] for r in range(world)]
