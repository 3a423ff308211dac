"""Training driver: builds the replica, the aggregation policy and the chief services, then runs
the step loop with the reference's log lines.

Mirrors `train()` of the reference drivers (reference: DistributedLeNet/src/distributed_train.py:105-373,
DistributedResNet/src/distributed_train.py:182-456, Distributed_test/src/distributed_train.py):
cluster sanity, chief election (worker 0), staircase learning rate whose decay period is divided
by the number of aggregated replicas, strategy switch on the method flags, startup barrier,
optional token wait + timing broadcasts for the worker-time CDF method, NaN guard, per-step
timeline dumps, the `Worker %d: %s: step %d, loss = %f(%.1f examples/sec; %.3f  sec/batch)` line,
stop when `step > max_steps`, `Elapsed Time` on the chief and a final checkpoint.
"""
from __future__ import annotations

import logging
import math
import os
import random
import sys
import time
from datetime import datetime
from typing import Dict, Optional


from .. import lr as lr_mod
from ..checkpoint.saver import Saver
from ..cluster import resolve_task
from ..data import cifar10 as cifar_data
from ..data import mnist as mnist_data
from ..parallel import control_plane
from ..parallel.comm import Communicator
from ..parallel.strategies import make_strategy
from .replica import make_replica
from .supervisor import Supervisor
from .timeline import StepTimeline

log = logging.getLogger("dtf")

TRAIN_LINE = "Worker %d: %s: step %d, loss = %f(%.1f examples/sec; %.3f  sec/batch)"


def setup_logging(level=logging.INFO) -> None:
    """tf.logging-like lines on stderr (the launcher redirects them to out_<ROLE_ID>)."""
    root = logging.getLogger("dtf")
    if not root.handlers:
        h = logging.StreamHandler(sys.stderr)
        h.setFormatter(logging.Formatter("%(levelname)s:dtf:%(message)s"))
        root.addHandler(h)
    root.setLevel(level)
    root.propagate = False


class _MnistFeed:
    def __init__(self, data_dir: str, batch_size: int):
        self.sets = mnist_data.load_mnist(data_dir)
        self.batch_size = batch_size
        self.num_examples = self.sets.train.num_examples

    def next(self):
        return self.sets.train.next_batch(self.batch_size)

    def to_gpu(self, device) -> None:
        """Moves the training set into HBM; from then on `feed_into` slices batches on the GPU."""
        self.gpu_batcher = mnist_data.GpuBatcher(self.sets.train, device)

    def feed_into(self, replica) -> None:
        st = replica.step
        with replica.stream_ctx():
            self.gpu_batcher.next_batch_into(st.images, st.labels)


class _CifarFeed:
    def __init__(self, data_dir: str, batch_size: int, padding: int, num_examples: int, seed: Optional[int]):
        data, labels = cifar_data.prepare_train_data(data_dir, padding)
        self.batcher = cifar_data.TrainBatcher(data, labels, padding, seed)
        self.batch_size = batch_size
        # the LR schedule uses the dataset size flag, not the array length (reference:
        # DistributedResNet/src/distributed_train.py:213-215)
        self.num_examples = num_examples if cifar_data.have_files(data_dir) else len(labels)

    def next(self):
        return self.batcher.next_batch(self.batch_size)

    def to_gpu(self, device) -> None:
        """Moves the padded dataset into HBM; from then on `feed_into` crops + whitens on the GPU."""
        b = self.batcher
        gb = cifar_data.GpuTrainBatcher(b.data, b.labels, b.padding, None, device)
        gb.rng = b.rng   # continue the same random stream
        self.gpu_batcher = gb

    def feed_into(self, replica) -> None:
        st = replica.step
        with replica.stream_ctx():
            self.gpu_batcher.next_batch_into(st.images, st.labels)


_T0 = time.time()


def stage(rank: int, what: str) -> None:
    """Timestamped bring-up marker (stderr, every rank): when a multi-GPU job stops during start-up,
    the last marker of each rank names the layer it stopped in."""
    log.info("STAGE worker=%d +%.3fs %s", rank, time.time() - _T0, what)


def _inject_straggler(flags, rank: int, rng: random.Random) -> None:
    if flags.inject_straggler_ms <= 0 or not flags.inject_straggler_ranks:
        return
    ranks = {int(r) for r in flags.inject_straggler_ranks.split(",") if r.strip()}
    if rank in ranks and rng.random() < flags.inject_straggler_prob:
        time.sleep(flags.inject_straggler_ms / 1e3)


def ps_main(flags) -> int:
    """`--job_name=ps`: nothing to host (parameters live in the workers' HBM). Waits for the
    workers' job to end so reference launch scripts that start a ps process keep working."""
    log.info("ps task %d: parameters are resident on the worker GPUs; idling until the job ends", flags.task_id)
    name = flags.board_name
    if not name:
        return 0
    from .. import _host

    try:
        n_workers = max(1, len([h for h in flags.worker_hosts.split(",") if h.strip()]))
        board = _host.Board(name, n_workers, False)
    except RuntimeError:
        return 0
    while not board.should_stop():
        time.sleep(0.2)
    return 0


def train(flags, model: str) -> Dict:
    """Runs one worker task to completion; returns a summary dict (steps, final loss, ...)."""
    setup_logging()
    task = resolve_task(flags.job_name, flags.task_id, flags.ps_hosts, flags.worker_hosts)
    if task.job_name == "ps":
        ps_main(flags)
        return {"role": "ps"}
    task.export()
    device = flags.device
    comm = Communicator.from_env(backend=flags.comm_backend or None, device="cpu" if device == "cpu" else None)
    world, rank = comm.world, comm.rank
    stage(rank, "process group up: %d rank(s), backend %s" % (world, comm.topology().get("backend", "?") if hasattr(comm, "topology") else "?"))
    is_chief = rank == 0
    num_workers = world
    # Both must be > 0 in a distributed run (reference: distributed_train.py:120-122); the ps
    # count is only checked for shape since no ps process holds state here.
    assert num_workers > 0, "num_workers must be > 0"
    num_replicas_to_aggregate = num_workers if flags.num_replicas_to_aggregate == -1 else flags.num_replicas_to_aggregate

    train_dir = flags.train_dir
    if is_chief:
        os.makedirs(train_dir, exist_ok=True)

    # ---- control board
    # One node: POSIX shared memory (csrc/host/board.cc), and the update log / asynchronous shards map
    # peers' memory. Ranks on several machines: the same board interface over the process group's
    # key-value store (parallel/store_board.py), with the host update log's regions as store values -
    # the straggler policies then work across machines the way the reference's do through its
    # parameter server (the sharded optimizer is collectives plus a checkpoint rendezvous on the board,
    # so it runs there too). What still needs one node: the asynchronous shards, which move data
    # through mapped memory.
    needs_board = world > 1
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    spans_machines = world > 1 and local_world < world
    plane = getattr(flags, "control_plane", "auto") or "auto"
    if plane == "auto":
        plane = os.environ.get("DTF_CONTROL_PLANE") or ("store" if spans_machines else "shm")
    if plane not in ("shm", "store"):
        raise ValueError("--control_plane must be auto, shm or store")
    if spans_machines and plane == "shm":
        raise ValueError(f"{world} ranks over machines of {local_world}: the shared-memory control board ends at the machine; "
                         "use --control_plane=store (or auto)")
    if world > 1 and plane == "store":
        if flags.async_method:
            raise ValueError("--async_method moves data through mapped memory and runs within one machine "
                             "(--control_plane=shm); the other aggregation policies run over the store")
        if world > 16:
            wants_log = [f for f in ("interval_method", "soft_kill_method", "backup_worker_method", "weighted_grads_method",
                                     "worker_times_cdf_method") if getattr(flags, f, False)]
            if wants_log:
                raise ValueError(f"--{wants_log[0]} over the store-backed control board serves at most 16 workers")
            # plain synchronous training only needs the process group
            log.info("Worker %d: %d ranks: synchronous all-reduce over the process group, no control board", rank, world)
            needs_board = False
        else:
            log.info("Worker %d: %d ranks coordinate through the process group's key-value store", rank, world)
    board = None
    if needs_board:
        name = flags.board_name or control_plane.board_name_for(train_dir, task.master_port)
        board = control_plane.open_board(name, world, rank, control_plane.run_nonce(comm), kind=plane)
        if is_chief and plane == "shm":
            control_plane.unlink_on_sigterm(name)

    # ---- data
    batch_size = flags.batch_size
    if model == "lenet":
        feed = _MnistFeed("" if flags.synthetic_data else flags.data_dir, batch_size)
    else:
        feed = _CifarFeed("" if flags.synthetic_data else flags.data_dir, batch_size, flags.padding_size,
                          flags.num_of_instances_cifar10, flags.random_seed + rank)
    decay_steps = lr_mod.decay_steps(feed.num_examples, batch_size, flags.num_epochs_per_decay, num_replicas_to_aggregate)

    # ---- replica + strategy
    uses_log = flags.interval_method or flags.soft_kill_method or flags.backup_worker_method or \
        getattr(flags, "weighted_grads_method", False) or flags.async_method
    if flags.grad_buckets > 0:
        buckets = flags.grad_buckets
    elif uses_log or world == 1:
        # the update log and the asynchronous shards take the whole gradient at the end of the pass
        # (short-circuiting needs no segment boundaries: every kernel checks the abort word itself)
        buckets = 1
    else:
        from ..models.lenet import lenet_param_specs
        from ..models.resnet import resnet_param_specs
        from ..parallel.comm import default_grad_buckets

        specs = lenet_param_specs() if model == "lenet" else resnet_param_specs(flags.num_residual_blocks)
        buckets = default_grad_buckets(sum(sp.numel for sp in specs))
    replica = make_replica(
        model, batch_size, device, num_residual_blocks=flags.num_residual_blocks, optimizer=flags.optimizer,
        momentum=flags.momentum, weight_decay=flags.weight_decay if flags.optimizer == "momentum" else 0.0,
        num_buckets=buckets, use_graph=flags.hip_graph,
        # kernels that return at once when the update they work for is complete (GPU replicas)
        abortable=bool(world > 1 and (flags.soft_kill_method or flags.backup_worker_method)),
        matmul=flags.matmul,
    )
    # On the GPU the dataset lives in HBM: ResNet crops + whitens with one kernel per step, LeNet
    # slices its batches device-to-device
    stage(rank, "replica built (%s)" % ("step program recorded, hipGraph instantiated" if getattr(replica, "use_graph", False)
                                        else "step program recorded" if hasattr(replica, "step") else "CPU"))
    gpu_feed = hasattr(replica, "step") and flags.gpu_data_pipeline
    if gpu_feed:
        feed.to_gpu(replica.device)
    saver = Saver(replica.arena, flags.optimizer, flags.checkpoint_names)
    sv = Supervisor(saver, train_dir, is_chief, flags.save_interval_secs)
    global_step = 0
    restored = sv.restore_latest()
    if restored is not None:
        global_step = restored
        replica.updates_applied = restored
    # non-chief replicas take the chief's state (reference: prepare_or_wait_for_session)
    with replica.stream_ctx():
        comm.broadcast(replica.arena.params, 0)
        for t in replica.arena.slots.values():
            comm.broadcast(t, 0)
    replica.synchronize()
    if world > 1:
        stage(rank, "parameters broadcast (first collective done)")

    if flags.allreduce_impl == "p2p":
        comm.enable_p2p(replica.arena.grads_full.numel())
        stage(rank, "peer handles imported (p2p all-reduce)")
    elif flags.allreduce_impl != "rccl":
        raise ValueError("--allreduce_impl must be rccl or p2p")

    async_mode = flags.async_method
    if async_mode:
        from ..parallel.async_ps import AsyncParameterShards

        name = (flags.board_name or control_plane.board_name_for(train_dir, task.master_port)) + "_ps"
        if board is None:
            board = control_plane.open_board(name + "_b", world, rank)
        shards = AsyncParameterShards(
            replica, comm, board, flags.staleness_bound if flags.staleness_bound >= 0 else 2 * world, name,
            lr_mod.StaircaseSchedule(flags.initial_learning_rate, decay_steps, flags.learning_rate_decay_factor),
            # every owner's shard continues at the restored step: the learning-rate staircase and
            # Adam's bias correction are functions of the shard's version
            start_version=global_step,
        )
        board.set_global_step(global_step) if is_chief else None
        stage(rank, "peer handles imported (asynchronous shards)")
        strategy = None
    else:
        strategy = make_strategy(flags, comm, board, num_replicas_to_aggregate)
        strategy.start_version = global_step
        if strategy.uses_update_log and world > 1 and board is not None:
            ulog = strategy._get_log(replica)
            stage(rank, "update log ready (%s)" % type(ulog).__name__)
            # nobody offers a gradient for an update beyond the end of the run, so a worker that reads
            # the (lagging) counters late cannot push the global step past max_steps + 1
            ulog.max_version = flags.max_steps + 1
            if hasattr(ulog, "sync_status"):
                # with a loss read-back every step the stream is synchronised anyway; otherwise the
                # host only enqueues and reads the counters the kernels mirror into pinned memory
                ulog.sync_status = flags.log_every <= 1
        log.info("Worker %d: aggregation policy %s over %d replicas (aggregate %d)", rank, strategy.name, world, num_replicas_to_aggregate)

    timeout_client, timeout_server = control_plane.launch_manager(board, rank, world, flags.collective_timeout_s)
    if board is None and world > 1:
        comm.barrier()      # the ready barrier, through the process group
    watchdog = None
    if board is not None and world > 1:
        silence = flags.liveness_timeout_s if flags.liveness_timeout_s > 0 else flags.collective_timeout_s / 4.0

        def tolerate(dead) -> bool:
            # policies on the update log go on while enough replicas are left to make an update
            if strategy is None or not strategy.uses_update_log:
                return False
            if strategy.name == "backup":
                return world - len(dead) >= strategy.m
            if strategy.name == "interval":
                return 0 not in dead       # the chief's timer closes the windows
            return False

        def on_dead(r: int) -> None:
            lg = getattr(strategy, "log", None)
            if lg is not None and hasattr(lg, "mark_dead"):
                lg.mark_dead(r)

        watchdog = control_plane.Watchdog(board, rank, world, silence, tolerate, on_dead)
        watchdog.start()
    timeline = StepTimeline(rank, enabled=flags.timeline_logging)
    if flags.timeline_logging:
        replica.timeline = timeline        # traced steps: one event per kernel, phase spans from the policy
    summary_writer = None
    if is_chief and flags.should_summarize:
        from ..utils.summary import FileWriter

        summary_writer = FileWriter(train_dir)
    next_summary_time = time.time() + flags.save_summaries_secs
    begin_time = time.time()
    cur_iteration = -1
    rng = random.Random(flags.random_seed * 1000 + rank)
    loss_value = float("nan")
    cdf = flags.worker_times_cdf_method
    if cdf and board is not None and is_chief:
        for r in range(world):
            board.put_tokens(r, 1, global_step)   # init tokens (reference: get_init_tokens_op)
    aborted_steps = excluded_steps = 0
    lr_schedule = lr_mod.StaircaseSchedule(flags.initial_learning_rate, decay_steps, flags.learning_rate_decay_factor)
    sharded = (not async_mode) and getattr(strategy, "reduce_scatter", False) and world > 1
    window_start, window_steps = time.time(), 0   # since the last training line (log_every > 1)

    while not sv.should_stop():
        log.info("A new iteration...")
        cur_iteration += 1
        if cdf:
            if board is not None:
                tok = board.take_token(rank, flags.collective_timeout_s)   # opt._wait_op
                if tok is None:
                    break
            timeout_client.broadcast_worker_dequeued_token(cur_iteration)
        start_time = time.time()
        if gpu_feed:
            feed.feed_into(replica)
        else:
            images, labels = feed.next()
            replica.load_batch(images, labels)
        _inject_straggler(flags, rank, rng)
        timeline.reset(getattr(replica, "stream", None))
        log.info("RUNNING SESSION... %f" % time.time())
        if async_mode:
            try:
                with timeline.span("async_step"):
                    loss_value, step = shards.step()
            except RuntimeError as e:
                if board.failed_rank() == rank:
                    log.error("Worker %d: %s", rank, e)
                raise
            if board.global_step() > flags.max_steps:
                board.request_stop()
        else:
            try:
                with timeline.span("train_step", getattr(replica, "stream", None)):
                    res = strategy.step(replica, cur_iteration, lr_schedule)
            except RuntimeError as e:
                # A collective that lost its peer fails at once on some transports (TCP resets) and
                # hangs on others; either way the liveness watchdog is the one that says WHO died and
                # ends the process. Give it its time before passing the transport's error on.
                if board is not None and board.failed_rank() == rank:
                    log.error("Worker %d: %s", rank, e)     # this worker's own helper thread: nothing to wait for
                    raise
                if watchdog is not None:
                    log.error("Worker %d: step failed (%s); waiting for the liveness verdict", rank, str(e).splitlines()[0][:200])
                    watchdog.join(timeout=2.0 * watchdog.silence_s + 5.0)
                raise
            if res.global_step is not None:
                global_step = res.global_step     # a replica that catches up applies several updates in one call
            elif res.applied:
                global_step += 1
            aborted_steps += int(res.aborted)
            excluded_steps += int(not res.included)
            # per-step aggregator record: what the reference observes with tf.Print on the
            # parameter server (accumulated count, local vs global step, who was dropped;
            # reference: DistributedLeNet/src/soft_kill_optimizer.py:425-442,465-469)
            log.debug("AGGREGATOR worker=%d iter=%d policy=%s applied=%s contributors=%d included=%s aborted=%s "
                      "local_step=%d global_step=%d compute_s=%.6f", rank, cur_iteration, strategy.name, res.applied,
                      res.contributors, res.included, res.aborted, cur_iteration, global_step, res.compute_s)
            step = global_step
            if board is not None and is_chief:
                board.set_global_step(global_step)
        log.info("DONE RUNNING SESSION...")
        if cur_iteration == 0:
            replica.synchronize()
            stage(rank, "first step done")
        if cdf:
            timeout_client.broadcast_worker_finished_computing_gradients(cur_iteration)
            if is_chief:
                deadline = time.time() + 5.0
                while not timeout_server.collect(cur_iteration) and time.time() < deadline:
                    time.sleep(0.0005)
                timeout_server.maybe_report(cur_iteration)
                if board is not None:
                    for r in range(world):
                        board.put_tokens(r, 1, global_step)
                timeout_client.broadcast_parameters_updated(global_step)

        log_now = flags.log_every <= 1 or cur_iteration % flags.log_every == 0
        if log_now and not async_mode:
            loss_value = replica.loss_value()
            comm.check_health()          # the stream is synchronised here anyway
        if log_now:
            assert not math.isnan(loss_value), "Model diverged with loss = NaN"
        if flags.timeline_logging:
            timeline.write(train_dir, step)
        if step > flags.max_steps or (async_mode and board.should_stop()):
            break
        window_steps += 1
        if log_now:
            # With log_every > 1 the host runs ahead of the GPU between lines and the loss
            # read-back above waits for all of it, so the rate is taken over the whole window.
            now = time.time()
            duration = (now - window_start) / window_steps if flags.log_every > 1 else now - start_time
            window_start, window_steps = now, 0
            examples_per_sec = batch_size / float(duration)
            log.info(TRAIN_LINE % (rank, datetime.now(), step, loss_value, examples_per_sec, duration))
        if is_chief and summary_writer is not None and next_summary_time < time.time():
            log.info("Running Summary operation on the chief.")
            scalars = {"loss": loss_value, "learning_rate": lr_mod.exponential_decay(
                flags.initial_learning_rate, global_step, decay_steps, flags.learning_rate_decay_factor)}
            # activation sparsity of every conv output (reference: activation_summary,
            # DistributedResNet/src/resnet.py:13-20)
            scalars.update(getattr(getattr(replica, "step", None), "activation_sparsity", dict)())
            summary_writer.add_scalars(scalars, step)
            histograms = getattr(getattr(replica, "step", None), "activation_histograms", dict)()
            if histograms:
                summary_writer.add_histograms(histograms, step)
            log.info("Finished running Summary operation.")
            next_summary_time += flags.save_summaries_secs
        if sharded:
            # each range of the optimizer slots is current only on its owner: every replica joins
            # the gather at the SAME iteration, which the chief announces two iterations ahead
            if is_chief and sv.save_due(step) and board.save_step() < cur_iteration:
                board.request_save(cur_iteration + 2)
            if board.save_step() == cur_iteration:
                strategy.gather_optimizer_state(replica)
                sv.maybe_save(step, replica.updates_applied, force=True)
        else:
            if async_mode and sv.save_due(step):
                # the chief copies every owner's authoritative parameters and optimizer slots out of
                # the mapped shards: the checkpoint holds all variables, not just the chief's range
                shards.gather_state()
            sv.maybe_save(step, replica.updates_applied)
    lost = board is not None and board.agg_dead() != 0   # workers the watchdog gave up on: no collective can include them

    if async_mode:
        shards.drain(flags.collective_timeout_s)
        global_step = board.global_step()
        shards.close()
    if strategy is not None:
        strategy.finish(collective=not lost)
        if strategy.uses_update_log:
            global_step = replica.updates_applied
        if sharded:
            strategy.gather_optimizer_state(replica)
    replica.synchronize()
    if is_chief:
        log.info("Elapsed Time: %f" % (time.time() - begin_time))
    sv.stop()
    final = sv.final_save(global_step, replica.updates_applied)
    if summary_writer is not None:
        summary_writer.close()
    if watchdog is not None:
        watchdog.depart()
    if not lost:
        comm.barrier()
    if board is not None and is_chief:
        board.request_stop()
    out = {
        "role": "worker", "rank": rank, "world": world, "global_step": global_step, "iterations": cur_iteration + 1,
        "loss": loss_value, "checkpoint": final, "aborted_steps": aborted_steps, "excluded_steps": excluded_steps,
        "strategy": "async" if async_mode else strategy.name,
    }
    if async_mode:
        out["async_stats"] = dict(vars(shards.stats), staleness_histogram=list(shards.staleness_hist))
    if lost:
        out["dead_workers"] = [r for r in range(world) if (board.agg_dead() >> r) & 1]
    else:
        comm.close()      # tearing the process group down would wait for the dead worker
    import json

    log.info("RESULT %s", json.dumps(out, default=str))
    return out
