#!/usr/bin/env python3
"""Headline benchmark: images/sec of synchronous data-parallel ResNet CIFAR-10 training.

  python bench.py --gpus N --steps K --warmup W [--model resnet20|lenet|...] [--batch 256] [--input replay|gpu_batcher]

N > 1 runs one rank per GPU over RCCL. Under torchrun RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*
come from the environment; started plainly as `python bench.py --gpus N` the script starts the N
ranks itself (fresh child processes, created before anything touches the GPU) and relays rank 0's
JSON line. It refuses to run when the number of ranks differs from --gpus, or when fewer GPUs than
ranks are visible (unless --share-gpu asks for a rehearsal of several ranks on one GPU over gloo, in
which case `n_gpus` reports the distinct GPUs, not the ranks). Every rank trains the same pre-activation ResNet-(6n+2) on
synthetic CIFAR-shaped data with random-init weights: full forward, backward, gradient
all-reduce (bucketed, overlapped with backward) and the fused Adam update, all inside the timed
region, in fp32 (the reference's dtype) on the exact-fp32 matrix cores. Scaling is weak: the
per-GPU batch is fixed. Rank 0 prints one JSON line.

Reference workload: DistributedResNet/src/distributed_train.py (sync-replica Adam training of
the CIFAR ResNet; its launcher uses batch 256 per worker, DistributedResNet/tools/tf_ec2.py:88).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from distributed_tensorflow_amd import launch  # noqa: E402

MODELS = {"resnet8": 1, "resnet20": 3, "resnet32": 5, "resnet56": 9, "resnet110": 18, "lenet": 0}
T_START = time.time()


def mark(stage: str) -> None:
    """Timestamped per-rank stage marker on stderr: when a multi-GPU run stops, the last marker of
    every rank says which layer it stopped in (rendezvous, peer mapping, graph capture, first step,
    first collective)."""
    print("[bench rank %s +%.3fs] %s" % (os.environ.get("RANK", "0"), time.time() - T_START, stage), file=sys.stderr, flush=True)


def cpu_floor(model: str):
    try:
        with open(os.path.join(ROOT, "bench", "cpu_floor.json")) as f:
            d = json.load(f)
        if d.get("model") == model:
            return float(d["images_per_sec"])
    except (OSError, ValueError, KeyError):
        pass
    return None


def check_topology(args, comm) -> dict:
    """Refuses to measure something other than what --gpus asked for; returns the fields that say
    what the ranks really ran on (backend, ranks, distinct GPUs, ranks that talk through RCCL)."""
    if comm.world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {comm.world} rank(s); start it with "
                         f"`python bench.py --gpus {args.gpus}` (it launches the ranks itself) or under torchrun with "
                         f"--nproc-per-node {args.gpus}")
    topo = comm.topology()
    if topo["distinct_gpus"] < comm.world and not args.share_gpu:
        raise SystemExit(f"bench.py: {comm.world} ranks but only {topo['distinct_gpus']} distinct GPU(s); "
                         "pass --share-gpu (with --backend gloo) to rehearse several ranks on one GPU")
    return topo


def bench_policy(args) -> None:
    """Images/sec of the non-headline aggregation policies (same model, data, dtype and optimizer
    as the headline; each step is forward + backward + the policy's aggregation + update)."""
    from distributed_tensorflow_amd.parallel import control_plane
    from distributed_tensorflow_amd.parallel.comm import Communicator
    from distributed_tensorflow_amd.parallel.strategies import Backup, FullSync, Interval, SoftKill
    from distributed_tensorflow_amd.train.replica import HipReplica

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X")
    comm = Communicator.from_env(backend=args.backend, force_group=args.rehearse_collectives)
    world = comm.world
    topo = check_topology(args, comm)
    torch.manual_seed(1234 + comm.rank)
    replica = HipReplica("resnet", args.batch, num_residual_blocks=MODELS[args.model], optimizer=args.optimizer,
                         num_buckets=(args.buckets or 3) if world > 1 and args.mode in ("sharded", "async") else 1,
                         weight_decay=2e-4, abortable=world > 1 and args.mode in ("backup", "soft_kill"))
    with replica.stream_ctx():
        comm.broadcast(replica.arena.params, 0)
    board = None
    if args.mode in ("backup", "async", "interval", "soft_kill"):
        name = control_plane.board_name_for("bench_" + args.mode, int(os.environ.get("MASTER_PORT", "0")))
        if comm.is_chief:
            board = control_plane.open_board(name, world, 0)
        comm.barrier()
        if not comm.is_chief:
            board = control_plane.open_board(name, world, comm.rank)
    detail = {}
    if args.mode == "async":
        from distributed_tensorflow_amd.parallel.async_ps import AsyncParameterShards

        tau = args.staleness if args.staleness >= 0 else 2 * world
        detail["staleness_bound"] = tau
        shards = AsyncParameterShards(replica, comm, board, tau, name + "_ps", args.lr)
        shards.read_loss_every_step = False   # the host only enqueues: pull, step, push and the owners' applies are kernels
        step_fn = lambda i: shards.step()   # noqa: E731
    else:
        if args.mode == "backup":
            m = args.aggregate if args.aggregate > 0 else max(1, world - 1)
            strategy = Backup(comm, board, m)
            detail["replicas_to_aggregate"] = strategy.m
        elif args.mode == "interval":
            strategy = Interval(comm, board, args.interval_ms)
            detail["interval_ms"] = args.interval_ms
        elif args.mode == "soft_kill":
            strategy = SoftKill(comm, board, args.aggregate if args.aggregate > 0 else max(1, world - 1))
            detail["should_stop_worker_num"] = strategy.k
        else:
            strategy = FullSync(comm, board, reduce_scatter=True)
        if strategy.uses_update_log and world > 1:
            # the host never waits for the GPU inside the timed loop: the whole policy runs on the device
            strategy._get_log(replica).sync_status = False
        straggle_s = args.inject_straggler_ms / 1e3 if comm.rank == world - 1 and world > 1 else 0.0
        detail["inject_straggler_ms"] = args.inject_straggler_ms

        def step_fn(i):
            if straggle_s:
                time.sleep(straggle_s)     # the last rank is slow to even start its step
            return strategy.step(replica, i, args.lr)
    images = torch.randn(args.batch, 32, 32, 3, device=comm.device)
    labels = torch.randint(0, 10, (args.batch,), device=comm.device)
    replica.load_batch(images, labels)
    log_obj = strategy._get_log(replica) if args.mode != "async" and strategy.uses_update_log and world > 1 else None
    if log_obj is None:
        for i in range(args.warmup):
            step_fn(i)
        replica.synchronize()
        comm.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step_fn(args.warmup + i)
        replica.synchronize()
        torch.cuda.synchronize()
        comm.barrier()
        dt = time.perf_counter() - t0
        images_done = args.batch * world * args.steps
    else:
        # Update-log policies: replicas do not move in lockstep, so the run is bounded by UPDATES, like
        # the reference's max_steps on the global step: everybody iterates until `warmup` (untimed),
        # then until `warmup + steps` updates are applied; nobody contributes beyond that.
        it = 0

        def run_until(target: int) -> int:
            nonlocal it
            log_obj.max_version = target
            n = 0
            while log_obj.version < target:       # the mirror in pinned memory: no GPU synchronisation
                step_fn(it)
                it += 1
                n += 1
                if n % 8 == 0:
                    replica.synchronize()         # bound how far the host queues ahead of the counters it polls
            replica.synchronize()
            return n

        log_obj.sync_status = True
        run_until(args.warmup)
        log_obj.sync_status = False               # the host never waits for the GPU inside a timed step
        comm.barrier()
        torch.cuda.synchronize()
        acc0 = log_obj.contributed
        t0 = time.perf_counter()
        iters = run_until(args.warmup + args.steps)
        own_s = time.perf_counter() - t0
        torch.cuda.synchronize()
        comm.barrier()
        dt = time.perf_counter() - t0
        # per-rank view: how long an iteration took on THIS rank (a non-straggler must not pay for the
        # straggler) and what became of its gradients
        rows = comm.all_gather_object(dict(rank=comm.rank, iterations=iters, ms_per_iteration=round(own_s / max(1, iters) * 1e3, 4),
                                           version=log_obj.sync_version(), accepted=log_obj.contributed - acc0,
                                           dropped_total=log_obj.dropped))
        detail["per_rank"] = rows
        detail["updates_applied"] = args.steps
        detail["gradients_aggregated"] = sum(r["accepted"] for r in rows)
        detail["gradients_computed"] = sum(r["iterations"] for r in rows)
        images_done = args.batch * detail["gradients_aggregated"]   # images whose gradient entered an update
        strategy.finish()
    final_loss = replica.loss_value()
    if args.mode == "async":
        shards.drain(30.0)
        detail["async_stats"] = dict(vars(shards.stats), staleness_histogram=list(shards.staleness_hist), shard_versions=shards.versions())
    t = torch.tensor([dt], device=comm.device, dtype=torch.float64)
    if world > 1:
        import torch.distributed as dist

        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    if comm.is_chief:
        value = images_done / dt
        floor = cpu_floor(args.model)
        print(json.dumps({
            "metric": "images/sec (node) ResNet CIFAR-10 %s-SGD" % args.mode,
            "value": round(value, 1), "unit": "images/sec", "n_gpus": topo["distinct_gpus"], "ranks": world,
            "backend": topo["backend"], "distinct_gpus": topo["distinct_gpus"], "rccl_ranks": topo["rccl_ranks"],
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": round(value / floor, 2) if floor else None, "dtype": "fp32",
            "data": "synthetic CIFAR-10-shaped images, random-init weights",
            "config": dict(model=args.model, global_batch=args.batch * world, per_gpu_batch=args.batch, optimizer=args.optimizer,
                           parallelism=f"dp{world}", mode=args.mode, final_loss=round(final_loss, 4), **detail),
        }))
    if board is not None:
        comm.barrier()
        if args.mode == "async":
            shards.close()
        if comm.is_chief:
            type(board).unlink(name)
    comm.close()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--model", default="resnet20", choices=sorted(MODELS))
    ap.add_argument("--input", choices=["replay", "gpu_batcher"], default="replay",
                    help="replay: one device-resident batch is reused every step (the input pipeline is outside the timed "
                         "region). gpu_batcher: every timed step draws a fresh batch from a synthetic data set resident in "
                         "HBM through the training driver's own input path (ResNet: shuffled indices + random crop + "
                         "per-image whitening in one kernel, data/cifar10.py GpuTrainBatcher; LeNet: data/mnist.py GpuBatcher)")
    ap.add_argument("--matmul", choices=["fp32", "bf16x6"], default="fp32",
                    help="fp32 (the headline): every product on the exact-fp32 matrix instruction. bf16x6: opt-in second line - the "
                         "stride-1 ResNet layers (forward, and the fused backward kernels of the 16- and 32-channel stages) on the bf16 "
                         "matrix pipe, both operands split into three bf16 values, six partial products per product, fp32 accumulation (error against fp64 next to the exact "
                         "path's: profiles/r3_split_error_table.txt)")
    ap.add_argument("--no-second-line", action="store_true",
                    help="one-GPU ResNet runs with --matmul fp32 also time the bf16x6 path afterwards and report it as "
                         "\"second_line\" inside the JSON object (never as the headline value); this flag leaves that out")
    ap.add_argument("--batch", type=int, default=1024,
                    help="per-GPU batch (weak scaling). 1024 is sized for the MI355X (activations + gradients of ResNet-20 "
                         "take ~1.5 GB of the 288 GB HBM and each layer launches >=2 workgroups per CU); the reference's "
                         "launcher uses 256 per worker (DistributedResNet/tools/tf_ec2.py:88), reproduce with --batch 256")
    ap.add_argument("--optimizer", default="adam", choices=["adam", "momentum"])
    ap.add_argument("--buckets", type=int, default=0,
                    help="gradient buckets overlapped with backward (N>1); 0 = by gradient size (parallel/comm.py: default_grad_buckets)")
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto",
                    help="replay the recorded step as a hipGraph (on) or issue it from C++ on two streams (off); auto picks "
                         "the two-stream issue when the step uses a side stream: hipGraph replay serialises on ~10 us "
                         "cross-queue dependencies between the branches, measured slower (1.47 vs 1.24 ms/step)")
    ap.add_argument("--no-graph", action="store_true", help="same as --graph off")
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--backend", default=None, help="torch.distributed backend; default nccl (= RCCL). gloo lets several "
                                                     "ranks share one GPU to rehearse the multi-rank path")
    ap.add_argument("--allreduce", choices=["rccl", "p2p"], default="rccl",
                    help="gradient all-reduce: RCCL, or the hand-written two-shot kernel over peer-mapped HBM")
    ap.add_argument("--inject-straggler-ms", type=float, default=0.0,
                    help="policy modes: the last rank sleeps this long before every step")
    ap.add_argument("--interval-ms", type=float, default=5.0, help="interval mode: the chief closes a window this often")
    ap.add_argument("--mode", choices=["sync", "sharded", "backup", "interval", "soft_kill", "async"], default="sync",
                    help="sync = the headline (all-reduce of every replica's gradient). The others time the framework's "
                         "other aggregation policies through the same replica + strategy objects the training driver uses: "
                         "sharded = reduce-scatter + per-shard optimizer + all-gather; backup = mean of the first "
                         "--aggregate gradients (masked all-reduce); async = staleness-bounded parameter shards")
    ap.add_argument("--aggregate", type=int, default=0, help="backup mode: gradients per update (default N-1)")
    ap.add_argument("--staleness", type=int, default=-1,
                    help="async mode: staleness bound in shard versions (-1: twice the number of ranks - every rank pushes to "
                         "every shard once per iteration, so a version is already `ranks` behind after one round)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="allow more ranks than distinct GPUs (rehearsal of the multi-rank path on one GPU; needs --backend gloo, "
                         "RCCL refuses two ranks on one device). The JSON then reports n_gpus = distinct GPUs")
    ap.add_argument("--rehearse-collectives", action="store_true",
                    help="one rank only: create the process group anyway and run the bucketed step with its all-reduces "
                         "through the backend (exercises the multi-GPU code path of the step on a single GPU)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and not launch.launched_by_torchrun():
        # The parent never touches the GPU: it counts the devices in the kernel driver's sysfs tree
        # (launch.visible_gpus: no HIP runtime in this process), starts one fresh process per rank
        # and relays rank 0's JSON line. Every rank checks the count again from the inside
        # (check_topology: distinct GPUs behind the ranks).
        ndev = launch.visible_gpus()
        if ndev < args.gpus and not args.share_gpu:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but {ndev} GPU(s) visible; pass --share-gpu --backend gloo to "
                             "rehearse several ranks on one GPU")
        if args.share_gpu and ndev < args.gpus and (args.backend or "nccl") == "nccl":
            raise SystemExit("bench.py: --share-gpu needs --backend gloo (RCCL refuses two ranks on one device)")
        sys.stdout.flush()
        code = launch.spawn_local_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus, devices=ndev)
        raise SystemExit(code)
    if args.mode != "sync":
        if args.model == "lenet" or args.input != "replay":
            raise SystemExit("bench.py: the policy modes time ResNet steps on a replayed batch")
        return bench_policy(args)

    from distributed_tensorflow_amd.ops import reference as ref
    from distributed_tensorflow_amd.parallel.comm import Communicator

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; for the CPU floor run bench/cpu_reference.py")
    mark("start: creating the process group" if args.gpus > 1 or args.rehearse_collectives else "start")
    comm = Communicator.from_env(backend=args.backend, force_group=args.rehearse_collectives)
    world = comm.world
    multi = world > 1 or args.rehearse_collectives
    topo = check_topology(args, comm)
    if multi:
        mark("process group up: %d rank(s), backend %s, %d distinct GPU(s)" % (world, topo["backend"], topo["distinct_gpus"]))
    lenet = args.model == "lenet"
    n = MODELS[args.model]
    torch.manual_seed(1234 + comm.rank)
    if args.buckets <= 0:
        from distributed_tensorflow_amd.models.lenet import lenet_param_specs
        from distributed_tensorflow_amd.models.resnet import resnet_param_specs
        from distributed_tensorflow_amd.parallel.comm import default_grad_buckets

        args.buckets = default_grad_buckets(sum(sp.numel for sp in (lenet_param_specs() if lenet else resnet_param_specs(n))))

    def run_path(matmul: str) -> dict:
        """Builds the step program for one matrix path and times it: warm-up, then exactly --steps steps
        between barrier + synchronize on both sides."""
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            if lenet:
                from distributed_tensorflow_amd.models.lenet_hip import LeNetHipStep

                step = LeNetHipStep(args.batch, optimizer=args.optimizer, num_buckets=args.buckets if multi else 1, matmul=matmul)
            else:
                from distributed_tensorflow_amd.models.resnet_hip import ResNetHipStep

                step = ResNetHipStep(n, args.batch, optimizer=args.optimizer, num_buckets=args.buckets if multi else 1, matmul=matmul)
            mark("step program recorded: %d kernels, %d segment(s)" % (step.kernel_count, step.num_segments))
            # identical initial parameters on every replica
            comm.broadcast(step.arena.params, 0)
            if multi:
                mark("parameters broadcast (first collective done)")
            if args.allreduce == "p2p":
                comm.enable_p2p(step.arena.grads_full.numel())
                mark("peer handles imported (p2p all-reduce buffers mapped)")
            hw, ch = (28, 1) if lenet else (32, 3)
            images = torch.randn(args.batch, hw, hw, ch, device=comm.device)
            labels = torch.randint(0, 10, (args.batch,), device=comm.device)
            step.load_batch(images, labels)
            batcher = None
            if args.input == "gpu_batcher":
                import numpy as np

                rs = np.random.RandomState(99 + comm.rank)
                if lenet:
                    from distributed_tensorflow_amd.data import mnist as mnist_data

                    # MNIST-shaped synthetic training set (60000 x 28 x 28 x 1 in [-0.5, 0.5], int64 labels), resident in HBM
                    ds = mnist_data.DataSet(rs.uniform(-0.5, 0.5, size=(60000, 28, 28, 1)).astype(np.float32),
                                            rs.randint(0, 10, size=60000).astype(np.int64))
                    batcher = mnist_data.GpuBatcher(ds, comm.device)
                else:
                    from distributed_tensorflow_amd.data import cifar10 as cifar_data

                    # CIFAR-shaped synthetic training set, zero-padded by 2 like prepare_train_data's (50000 x 36 x 36 x 3)
                    data = np.zeros((50000, 36, 36, 3), dtype=np.float32)
                    data[:, 2:34, 2:34, :] = rs.randint(0, 256, size=(50000, 32, 32, 3)).astype(np.float32)
                    batcher = cifar_data.GpuTrainBatcher(data, rs.randint(0, 10, size=50000), 2, 7 + comm.rank, comm.device)
                    del data
                mark("synthetic data set resident in HBM")
            mode = "off" if args.no_graph else args.graph
            graph = (mode == "on") or (mode == "auto" and not getattr(step, "side_stream", False))
            if graph:
                step.instantiate(per_segment=multi)
                mark("hipGraph instantiated")
            nseg = step.num_segments
            ranges = step.bucket_ranges
            grads = step.arena.grads
            t_global = 0
            # events around the collectives of a step (N > 1): their share of the step says where a scaling loss comes from
            coll_events = []

            def one_step(timed: bool = False) -> None:
                nonlocal t_global
                t_global += 1
                if batcher is not None:
                    batcher.next_batch_into(step.images, step.labels)
                step.set_hyper(lr_t=ref.adam_lr_t(args.lr, t_global) if args.optimizer == "adam" else args.lr,
                               grad_scale=1.0 / world, momentum=0.9, weight_decay=2e-4, step=t_global if lenet else None)
                if not multi:
                    step.run(-1, graph)
                    return
                for i in range(nseg - 1):
                    step.run(i, graph)
                    lo, hi = ranges[i]
                    if i < nseg - 2:
                        comm.all_reduce_async(grads[lo:hi])   # overlaps the rest of backward
                    else:
                        # nothing left to overlap with: issued on the compute stream, so two events bracket exactly
                        # the part of the step that is exposed communication
                        if timed:
                            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            e0.record(stream)
                        comm.all_reduce_inline(grads[lo:hi])
                        if timed:
                            e1.record(stream)
                            coll_events.append((e0, e1))
                comm.wait_all()
                step.run(nseg - 1, graph)

            for w in range(args.warmup):
                one_step()
                if w == 0:
                    stream.synchronize()
                    mark("first step done" + (" (with its gradient all-reduce)" if multi else ""))
            stream.synchronize()
            comm.barrier()
            torch.cuda.synchronize()
            mark("warm-up done, timing %d steps" % args.steps)
            # one event per step boundary (recorded on the step's stream, never waited on inside the
            # timed region) gives the per-step GPU times for the p50
            marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
            t0 = time.perf_counter()
            marks[0].record(stream)
            for i in range(args.steps):
                one_step(timed=True)
                marks[i + 1].record(stream)
            stream.synchronize()
            torch.cuda.synchronize()
            comm.barrier()
            dt = time.perf_counter() - t0
            loss = step.loss_value()
            per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
            p50 = per_step[len(per_step) // 2] if per_step else float("nan")
            coll_ms = sum(e0.elapsed_time(e1) for e0, e1 in coll_events) / max(1, args.steps)
            mark("timed region done")
        return dict(dt=dt, p50=p50, loss=loss, coll_ms=coll_ms, graph=graph, buckets=len(ranges), kernels=step.kernel_count,
                    deterministic=bool(getattr(step, "deterministic", False)),
                    split=(len(getattr(step, "_split_fwd", ())), len(getattr(step, "_split_bwd", ()))))

    def max_over_ranks(dt: float) -> float:
        t = torch.tensor([dt], device=comm.device, dtype=torch.float64)
        if multi:
            import torch.distributed as dist

            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def matmul_label(matmul: str, split) -> str:
        if matmul == "fp32":
            return "fp32 (v_mfma_f32_16x16x4_f32)"
        if lenet:
            return "bf16x6 split, fp32 accumulate (second convolution: forward, data gradient and filter gradient; everything else fp32 MFMA)"
        return ("bf16x6 split, fp32 accumulate (stride-1 layers: %d forward and %d fused backward kernels; everything else "
                "fp32 MFMA)" % split)

    res = run_path(args.matmul)
    dt = max_over_ranks(res["dt"])
    # Second, separately labelled line inside the same JSON object (one GPU only, never the headline value):
    # the opt-in split-operand path on the same model, batch, steps and warm-up.
    second = None
    if args.matmul == "fp32" and world == 1 and not args.rehearse_collectives and not args.no_second_line:
        try:
            r2 = run_path("bf16x6")
            second = {"matmul": matmul_label("bf16x6", r2["split"]), "value": round(args.batch * args.steps / r2["dt"], 1),
                      "unit": "images/sec", "ms_per_step": round(r2["dt"] / args.steps * 1e3, 4), "p50_ms_per_step": round(r2["p50"], 4),
                      "final_loss": round(r2["loss"], 4), "kernels_per_step": r2["kernels"],
                      "error_table": None if lenet else "profiles/r3_split_error_table.txt (max error vs fp64 <= 1.33 x the exact path's)"}
        except Exception as e:   # noqa: BLE001 - the headline line must not depend on the opt-in path
            second = {"matmul": "bf16x6", "error": "%s: %s" % (type(e).__name__, e)}
    loss, p50, coll_ms, graph = res["loss"], res["p50"], res["coll_ms"], res["graph"]
    if comm.is_chief:
        ms = dt / args.steps * 1e3
        value = args.batch * world * args.steps / dt
        floor = cpu_floor(args.model)
        print(json.dumps({
            "metric": "images/sec (node) LeNet MNIST sync-SGD" if lenet else "images/sec (node) ResNet CIFAR-10 sync-SGD",
            "value": round(value, 1),
            "unit": "images/sec",
            "n_gpus": topo["distinct_gpus"],
            "ranks": world,
            "backend": topo["backend"],
            "distinct_gpus": topo["distinct_gpus"],
            "rccl_ranks": topo["rccl_ranks"],
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms, 4),
            "p50_ms_per_step": round(p50, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": round(value / floor, 2) if floor else None,
            "dtype": "fp32",
            "matmul": matmul_label(args.matmul, res["split"]),
            "data": "synthetic %s-shaped images, random-init weights" % ("MNIST" if lenet else "CIFAR-10"),
            "input": args.input,
            "config": {
                "model": args.model, "global_batch": args.batch * world, "per_gpu_batch": args.batch,
                "reference_worker_batch": 128 if lenet else 256, "seq_len": None, "image": "28x28x1" if lenet else "32x32x3",
                "optimizer": args.optimizer, "parallelism": f"dp{world}", "hip_graph": graph,
                "deterministic": res["deterministic"],
                "exposed_allreduce_ms_per_step": round(coll_ms, 4) if multi else None,
                "allreduce": ("p2p" if comm.p2p is not None else topo["backend"]) if multi else "none", "grad_buckets": res["buckets"],
                "kernels_per_step": res["kernels"], "final_loss": round(loss, 4),
                "peak_hbm_gb": round(torch.cuda.max_memory_allocated() / 2**30, 2),
                "baseline": "CPU reference floor (bench/cpu_floor.json); the reference publishes no numbers",
            },
            "second_line": second,
        }))
    comm.close()


def _run_with_clean_stdout() -> None:
    """Libraries below us may print to the process's stdout (RCCL prints a version banner there
    when NCCL_DEBUG is set); the contract is ONE JSON line on stdout. So file descriptor 1 points
    at stderr while the benchmark runs, and Python's sys.stdout keeps the real stdout."""
    sys.stdout.flush()
    real = os.dup(1)
    os.dup2(2, 1)
    sys.stdout = os.fdopen(real, "w", buffering=1)
    try:
        main()
    finally:
        sys.stdout.flush()


if __name__ == "__main__":
    _run_with_clean_stdout()
