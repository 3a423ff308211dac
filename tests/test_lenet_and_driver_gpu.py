"""GPU paths above the kernels: LeNet step program, the replica/strategy stack, the training
driver, and multi-process modes sharing the one available GPU."""
import os
import time
import re
import subprocess
import sys

import pytest
import torch

from mp_util import run_workers

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


@pytest.mark.parametrize("matmul", ["fp32", "bf16x6"])
def test_lenet_step_matches_autograd(matmul):
    from distributed_tensorflow_amd.models.lenet import lenet_loss
    from distributed_tensorflow_amd.models.lenet_hip import LeNetHipStep

    b = 12
    step = LeNetHipStep(b, dropout=False, num_buckets=2, matmul=matmul)
    g = torch.Generator().manual_seed(3)
    images = torch.rand(b, 28, 28, 1, generator=g) - 0.5
    labels = torch.randint(0, 10, (b,), generator=g)
    step.load_batch(images.cuda(), labels.cuda())
    step.set_hyper(lr_t=0.0)
    step.run(graph=False)
    torch.cuda.synchronize()
    params = {k: v.double().requires_grad_(True) for k, v in step.arena.state_dict().items()}
    loss = lenet_loss(params, images.double(), labels, train=False)
    grads = dict(zip(params, torch.autograd.grad(loss, list(params.values()))))
    assert abs(step.loss_value() - loss.item()) < 1e-4 * max(1, abs(loss.item()))
    for name, gref in grads.items():
        assert _rel(step.arena.grad(name), gref) < 2e-3, name
    assert step.bucket_ranges[0][0] == step.arena.offsets["fc1_weights"] and step.bucket_ranges[-1][0] == 0


def test_lenet_dropout_training_reduces_loss_under_graph_replay():
    from distributed_tensorflow_amd.models.lenet_hip import LeNetHipStep
    from distributed_tensorflow_amd.ops import reference as ref

    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        step = LeNetHipStep(64)
        g = torch.Generator().manual_seed(4)
        step.load_batch((torch.rand(64, 28, 28, 1, generator=g) - 0.5).cuda(), torch.randint(0, 10, (64,), generator=g).cuda())
        step.instantiate()
        losses = []
        for t in range(1, 31):
            step.set_hyper(lr_t=ref.adam_lr_t(1e-3, t))
            step.set_step_counter(t)
            step.run(graph=True)
            if t in (1, 30):
                s.synchronize()
                losses.append(step.loss_value())
    assert losses[1] < losses[0]


@pytest.mark.parametrize("matmul", ["fp32", "bf16x6"])
def test_driver_trains_resnet_on_gpu_and_evaluator_scores_it(tmp_path, matmul):
    d = str(tmp_path / "train")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "DistributedResNet/src/resnet_distributed_train.py", "--synthetic_data", "--batch_size=32",
                        "--num_residual_blocks=1", "--max_steps=12", f"--train_dir={d}", "--initial_learning_rate=0.001",
                        "--log_every=4", "--should_summarize=true", "--save_summaries_secs=0", f"--matmul={matmul}"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    steps = [int(m.group(1)) for m in (re.match(r".*step (\d+), loss = ", l) for l in r.stderr.splitlines()) if m]
    assert steps and steps[-1] >= 9
    # activation summaries (sparsity scalars + histograms) of every ReLU output, computed on the GPU
    import glob

    from distributed_tensorflow_amd.utils.summary import read_events, read_histograms

    (ev,) = glob.glob(os.path.join(d, "events.out.tfevents.*"))
    scalars = list(read_events(ev))
    assert scalars and "loss" in scalars[-1][1] and any(k.endswith("/sparsity") for k in scalars[-1][1])
    histos = list(read_histograms(ev))
    assert histos and "conv0/activations" in histos[-1][1]
    h = histos[-1][1]["conv0/activations"]
    assert h["num"] == 32 * 32 * 32 * 16 and h["min"] == 0.0 and sum(h["bucket"]) == h["num"]
    e = subprocess.run([sys.executable, "DistributedResNet/src/resnet_eval.py", "--synthetic_data", "--num_residual_blocks=1",
                        f"--checkpoint_dir={d}", f"--eval_dir={tmp_path / 'eval'}", "--run_once=true"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert e.returncode == 0, e.stderr[-3000:]
    assert "Succesfully loaded model from" in e.stderr and "Precision @ 1:" in e.stderr


def _gpu_sync_worker(rank, world, board_name, mode):
    # Several replicas share GPU 0; gloo carries the collectives (RCCL refuses two ranks on one device).
    import torch

    from distributed_tensorflow_amd.parallel import control_plane
    from distributed_tensorflow_amd.parallel.comm import Communicator
    from distributed_tensorflow_amd.parallel.strategies import Backup, FullSync, Interval, SoftKill, Weighted
    from distributed_tensorflow_amd.train.replica import HipReplica

    os.environ["LOCAL_RANK"] = "0"
    comm = Communicator.from_env(backend="gloo")
    board = control_plane.open_board(board_name, world, rank)
    rep = HipReplica("resnet", 16, num_residual_blocks=1, num_buckets=3, abortable=mode in ("backup", "soft_kill"))
    g = torch.Generator().manual_seed(50 + rank)
    rep.load_batch(torch.randn(16, 32, 32, 3, generator=g), torch.randint(0, 10, (16,), generator=g))
    control_plane.launch_manager(board, rank, world, 120)
    if mode == "p2p":
        comm.enable_p2p(rep.arena.grads_full.numel(), blocks=8)
    if mode == "backup":
        strat = Backup(comm, board, world - 1)
    elif mode == "interval":
        strat = Interval(comm, board, 5.0)          # the chief's timer closes a window every 5 ms
    elif mode == "soft_kill":
        strat = SoftKill(comm, board, 1)            # the first arrival stops the other replica's backward pass
    elif mode == "weighted":
        strat = Weighted(comm, board, 1, 142)       # one simulated kill per step
    else:
        strat = FullSync(comm, board, reduce_scatter=(mode == "sharded"))
    losses = []
    applied = 0
    for it in range(12 if mode == "interval" else 6):
        res = strat.step(rep, it, 1e-3)
        applied += int(res.applied)
        losses.append(rep.loss_value())
        if mode == "interval":
            time.sleep(0.004)
    strat.finish()
    rep.synchronize()
    psum = rep.arena.params.double().sum().item()
    comm.barrier()
    comm.close()
    return dict(losses=losses, psum=psum, updates=rep.updates_applied, applied=applied)


@pytest.mark.parametrize("mode", ["sync", "backup", "sharded", "p2p", "interval", "soft_kill", "weighted"])
def test_two_gpu_replicas_stay_identical(mode):
    out = run_workers(_gpu_sync_worker, 2, f"dtf_g_{mode}_{os.getpid()}", mode, timeout=300)
    assert abs(out[0]["psum"] - out[1]["psum"]) < 1e-6 * max(1.0, abs(out[0]["psum"]))
    if mode == "interval":
        # updates follow the timer, not the iteration count; both replicas end on the same one
        assert out[0]["updates"] == out[1]["updates"] >= 2
        return
    if mode == "backup":
        # one gradient per update (M = 1 of 2): every accepted gradient is an update of its own, so six
        # iterations per replica make between 6 and 12 updates, the same number on both
        assert out[0]["updates"] == out[1]["updates"] and 6 <= out[0]["updates"] <= 12
    else:
        assert all(o["updates"] == 6 for o in out)
    if mode in ("backup", "soft_kill", "weighted"):
        # the update may come from the OTHER replica's batch only, so a replica's own loss need not fall
        assert all(l == l and l < 10 for o in out for l in o["losses"])
    else:
        assert out[0]["losses"][-1] < out[0]["losses"][0]


def _gpu_async_worker(rank, world, board_name):
    import torch

    from distributed_tensorflow_amd.parallel import control_plane
    from distributed_tensorflow_amd.parallel.async_ps import AsyncParameterShards
    from distributed_tensorflow_amd.parallel.comm import Communicator
    from distributed_tensorflow_amd.train.replica import HipReplica

    os.environ["LOCAL_RANK"] = "0"
    comm = Communicator.from_env(backend="gloo")
    board = control_plane.open_board(board_name, world, rank)
    rep = HipReplica("resnet", 16, num_residual_blocks=1)
    g = torch.Generator().manual_seed(70 + rank)
    rep.load_batch(torch.randn(16, 32, 32, 3, generator=g), torch.randint(0, 10, (16,), generator=g))
    control_plane.launch_manager(board, rank, world, 120)
    shards = AsyncParameterShards(rep, comm, board, 8, board_name + "_ps", lambda v: 1e-3)
    losses = [shards.step()[0] for _ in range(8)]
    shards.drain(60)
    rep.synchronize()
    out = dict(losses=losses, psum=rep.arena.params.double().sum().item(), stats=vars(shards.stats),
               versions=shards.versions(), hist=list(shards.staleness_hist), kind=type(shards).__name__)
    shards.close()
    comm.barrier()
    comm.close()
    return out


def test_async_parameter_shards_in_hbm_over_ipc():
    out = run_workers(_gpu_async_worker, 2, f"dtf_g_async_{os.getpid()}", timeout=300)
    assert all(o["kind"] == "DeviceAsyncShards" for o in out)          # kernels over IPC-mapped HBM, not the host path
    assert out[0]["versions"] == [16, 16]
    assert sum(o["stats"]["applied"] for o in out) == 32
    assert all(sum(o["hist"]) == 16 and sum(o["hist"][9:]) == 0 for o in out)   # every push applied within the bound (8)
    assert abs(out[0]["psum"] - out[1]["psum"]) < 1e-6 * max(1.0, abs(out[0]["psum"]))
    assert out[0]["losses"][-1] < out[0]["losses"][0]


def _gpu_async_ckpt_worker(rank, world, board_name, tmp):
    import torch

    from distributed_tensorflow_amd.checkpoint.saver import Saver
    from distributed_tensorflow_amd.parallel import control_plane
    from distributed_tensorflow_amd.parallel.async_ps import AsyncParameterShards
    from distributed_tensorflow_amd.parallel.comm import Communicator
    from distributed_tensorflow_amd.train.replica import HipReplica

    os.environ["LOCAL_RANK"] = "0"
    comm = Communicator.from_env(backend="gloo")
    board = control_plane.open_board(board_name, world, rank)
    g = torch.Generator().manual_seed(70 + rank)
    images, labels = torch.randn(16, 32, 32, 3, generator=g), torch.randint(0, 10, (16,), generator=g)
    rep = HipReplica("resnet", 16, num_residual_blocks=1)
    rep.load_batch(images, labels)
    control_plane.launch_manager(board, rank, world, 120)
    shards = AsyncParameterShards(rep, comm, board, 8, board_name + "_ps", lambda v: 1e-3)
    for _ in range(6):
        shards.step()
    rep.synchronize()
    comm.barrier()
    shards.service()                    # nothing left in any inbox
    comm.barrier()
    own = {k: t.cpu() for k, t in shards.own_state().items()}
    versions = shards.versions()
    n = shards.shard_floats
    blank_outside = float(rep.arena.full("slot:m")[n:2 * n].abs().max()) if rank == 0 else 0.0
    prefix = None
    if rank == 0:
        shards.gather_state()           # peer reads of rank 1's shard and slots
        prefix = Saver(rep.arena, "adam").save(tmp, versions[0], versions[0])
    prefix = comm.all_gather_object(prefix)[0]
    shards.close()
    # resume in fresh replicas
    rep2 = HipReplica("resnet", 16, num_residual_blocks=1)
    rep2.load_batch(images, labels)
    step = Saver(rep2.arena, "adam").restore(prefix)
    shards2 = AsyncParameterShards(rep2, comm, board, 8, board_name + "_ps2", lambda v: 1e-3, start_version=step)
    back = {k: t.cpu() for k, t in shards2.own_state().items()}
    err = {k: float((back[k] - own[k]).abs().max()) for k in own}
    v0 = shards2.versions()
    comm.barrier()                      # nobody pushes before everybody has looked
    shards2.step()
    shards2.drain(60)
    v1 = shards2.versions()
    kind = type(shards2).__name__
    shards2.close()
    comm.barrier()
    comm.close()
    return dict(err=err, nonzero=float(own["m"].abs().sum()), versions=versions, step=step, v0=v0, v1=v1,
                blank_outside=blank_outside, kind=kind)


def test_async_checkpoint_holds_every_owners_optimizer_state_and_resumes(tmp_path):
    """Asynchronous run -> checkpoint written by the chief from the peer-mapped shards -> resume: every
    owner's m / v / parameters equal the pre-save values and the shard versions continue (the Saver of
    the reference stores all variables, Adam's slots included: DistributedLeNet/src/distributed_train.py:212,234-242)."""
    out = run_workers(_gpu_async_ckpt_worker, 2, f"dtf_g_asck_{os.getpid()}", str(tmp_path), timeout=300)
    assert all(o["kind"] == "DeviceAsyncShards" for o in out)
    assert out[0]["blank_outside"] == 0.0       # the chief's own arena never held rank 1's slots
    for o in out:
        assert o["nonzero"] > 0 and set(o["err"]) == {"params", "m", "v"} and all(e == 0.0 for e in o["err"].values()), o
        assert o["step"] == o["versions"][0] == 12
        assert o["v0"] == [12, 12] and o["v1"] == [14, 14]


@pytest.mark.parametrize("mode", ["backup", "backup_unsynced", "interval", "async", "soft_kill", "backup_host_fallback", "backup_store"])
def test_driver_runs_the_straggler_policies_on_the_device_update_log(tmp_path, mode):
    """Entry script -> driver -> policy, two workers sharing GPU 0 (gloo carries the set-up exchange, the
    aggregation itself runs in kernels over IPC-mapped HBM): both reach max_steps, the chief's
    checkpoint carries the final global step and worker 1 (slowed down) is the one whose gradients
    are dropped in backup mode. backup_host_fallback: rank 1 cannot map its peer's buffers; both ranks
    must notice together and run the same protocol through the host update log."""
    import json

    from distributed_tensorflow_amd.checkpoint.saver import latest_checkpoint, read_bundle
    from distributed_tensorflow_amd.launch import free_port, rank_environment

    d = str(tmp_path)
    extra = {"backup": ["--backup_worker_method=true", "--num_replicas_to_aggregate=1"],
             # loss read back every 4th iteration only: the host never waits for the GPU in between
             "backup_unsynced": ["--backup_worker_method=true", "--num_replicas_to_aggregate=1", "--log_every=4"],
             "interval": ["--interval_method=true", "--interval_ms=3"],
             "async": ["--async_method=true", "--staleness_bound=8"],
             "soft_kill": ["--soft_kill_method=true", "--should_stop_worker_num=1"],
             "backup_host_fallback": ["--backup_worker_method=true", "--num_replicas_to_aggregate=1"],
             # GPU replicas whose workers coordinate as if on different machines: gradients staged through the host and the store
             "backup_store": ["--backup_worker_method=true", "--num_replicas_to_aggregate=1", "--control_plane=store"]}[mode]
    port = free_port()
    procs, logs = [], []
    base_env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if mode == "backup_host_fallback":
        base_env["DTF_ULOG_FAIL_IMPORT"] = "1"
    for r in range(2):
        env = rank_environment(r, 2, port, 1, base_env)
        logf = open(os.path.join(d, f"rank{r}.log"), "w")
        logs.append(logf)
        procs.append(subprocess.Popen(
            [sys.executable, "DistributedResNet/src/resnet_distributed_train.py", "--synthetic_data", "--batch_size=32",
             "--num_residual_blocks=1", "--max_steps=30", f"--train_dir={d}", "--initial_learning_rate=0.001",
             "--comm_backend=gloo", "--save_interval_secs=100000", "--inject_straggler_ms=5", "--inject_straggler_ranks=1"] + extra,
            stdout=logf, stderr=subprocess.STDOUT, env=env, cwd=ROOT))
    try:
        rcs = [p.wait(timeout=280) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for f in logs:
            f.close()
    out = [open(os.path.join(d, f"rank{r}.log")).read() for r in range(2)]
    assert rcs == [0, 0], (rcs, out[0][-2500:], out[1][-2500:])
    ck = read_bundle(latest_checkpoint(d))
    assert int(ck["global_step"]) >= 30
    results = [json.loads(l.split("RESULT ", 1)[1]) for o in out for l in o.splitlines() if "RESULT {" in l]
    assert len(results) == 2
    # a worker that catches up applies several updates per iteration, so its log lines skip steps:
    # what every worker must reach is the global step
    assert all(r["global_step"] > 30 for r in results), results
    steps0 = [int(m.group(1)) for m in (re.match(r".*step (\d+), loss = ", l) for l in out[0].splitlines()) if m]
    # the last line a worker logs is the last global step it SAW before max_steps: pushes of the other
    # worker (async) or a catch-up of several updates can carry it past the final steps unlogged
    assert steps0 and steps0[-1] >= (20 if mode == "backup_unsynced" else 24)
    if mode == "backup":
        by_rank = {r["rank"]: r for r in results}
        assert by_rank[1]["excluded_steps"] > by_rank[0]["excluded_steps"]
    if mode == "backup_host_fallback":
        assert all("using the host update log" in o for o in out), (out[0][-1500:], out[1][-1500:])


def test_backup_workers_go_on_when_a_gpu_worker_is_killed(tmp_path):
    """Three workers share GPU 0 under backup workers (2 of 3); worker 2 is SIGKILLed mid-run. The
    survivors' watchdogs name it, take it out of the update log's back-pressure (and stand in for
    any shard it owned), and both reach max_steps."""
    import signal

    from distributed_tensorflow_amd.launch import free_port, rank_environment

    d = str(tmp_path)
    port = free_port()
    procs, logs = [], []
    for r in range(3):
        env = rank_environment(r, 3, port, 1, dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
        logf = open(os.path.join(d, f"rank{r}.log"), "w")
        logs.append(logf)
        procs.append(subprocess.Popen(
            [sys.executable, "DistributedResNet/src/resnet_distributed_train.py", "--synthetic_data", "--batch_size=32",
             "--num_residual_blocks=1", "--max_steps=400", f"--train_dir={d}", "--initial_learning_rate=0.001",
             "--comm_backend=gloo", "--save_interval_secs=100000", "--liveness_timeout_s=2", "--inject_straggler_ms=10",
             "--inject_straggler_ranks=0,1,2", "--backup_worker_method=true", "--num_replicas_to_aggregate=2"],
            stdout=logf, stderr=subprocess.STDOUT, env=env, cwd=ROOT))
    try:
        deadline = time.time() + 120
        seen = False
        while time.time() < deadline and not seen:
            steps = [int(m.group(1)) for m in (re.search(r"step (\d+), loss = ", l) for l in open(os.path.join(d, "rank2.log"))) if m]
            seen = bool(steps) and steps[-1] >= 20
            time.sleep(0.1)
        assert seen, open(os.path.join(d, "rank2.log")).read()[-2000:]
        procs[2].send_signal(signal.SIGKILL)
        rcs = [procs[r].wait(timeout=200) for r in (0, 1)]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for f in logs:
            f.close()
    out = [open(os.path.join(d, f"rank{r}.log")).read() for r in range(2)]
    assert rcs == [0, 0], (rcs, out[0][-2500:], out[1][-2500:])
    assert all("worker 2 has been silent" in o and "continuing without it" in o for o in out)
    import json

    results = [json.loads(l.split("RESULT ", 1)[1]) for o in out for l in o.splitlines() if "RESULT {" in l]
    assert len(results) == 2 and all(r["global_step"] > 400 and r["dead_workers"] == [2] for r in results), results
