"""Fused ResNet step on the GPU against the plain PyTorch autograd model of the same network."""
import os

import pytest
import torch

from distributed_tensorflow_amd.models.resnet import resnet_loss, resnet_param_specs
from distributed_tensorflow_amd.ops import reference as ref
from distributed_tensorflow_amd.runtime.arena import ParamArena

pytestmark = pytest.mark.gpu


def _make(n, batch, optimizer="adam", num_buckets=1, matmul=None):
    from distributed_tensorflow_amd.models.resnet_hip import ResNetHipStep

    step = ResNetHipStep(n, batch, optimizer=optimizer, num_buckets=num_buckets, matmul=matmul)
    g = torch.Generator().manual_seed(1234)
    # non-trivial BN affine parameters so their gradients are exercised
    for s in step.arena.specs:
        if s.name.endswith("gamma"):
            step.arena.param(s.name).copy_(1.0 + 0.2 * torch.randn(s.shape, generator=g))
        if s.name.endswith("beta"):
            step.arena.param(s.name).copy_(0.1 * torch.randn(s.shape, generator=g))
    images = torch.randn(batch, 32, 32, 3, generator=g)
    labels = torch.randint(0, 10, (batch,), generator=g)
    step.load_batch(images.cuda(), labels.cuda())
    return step, images, labels


def _reference_grads(step, images, labels, n, dtype=torch.float64):
    params = {k: v.to(dtype).requires_grad_(True) for k, v in step.arena.state_dict().items()}
    loss = resnet_loss(params, images.to(dtype), labels, n)
    grads = torch.autograd.grad(loss, list(params.values()))
    return loss.item(), dict(zip(params.keys(), grads)), params


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


@pytest.mark.parametrize("matmul", ["fp32", "bf16x6"])
@pytest.mark.parametrize("n,num_buckets", [(1, 1), (3, 3)])
def test_fused_step_matches_autograd(n, num_buckets, matmul):
    step, images, labels = _make(n, 8, num_buckets=num_buckets, matmul=matmul)
    step.set_hyper(lr_t=0.0)
    step.run(graph=False)
    torch.cuda.synchronize()
    loss_ref, grads_ref, _ = _reference_grads(step, images, labels, n)
    # fp32 PyTorch against fp64 PyTorch calibrates how much rounding alone moves each gradient
    _, grads32, _ = _reference_grads(step, images, labels, n, torch.float32)
    assert abs(step.loss_value() - loss_ref) < 2e-4 * max(1.0, abs(loss_ref))
    for name, gref in grads_ref.items():
        got = step.arena.grad(name)
        err = _rel(got, gref)
        noise = _rel(grads32[name], gref)
        assert err < max(2e-3, 4.0 * noise), f"{name}: gradient relative error {err:.2e} (fp32 noise {noise:.2e})"
    if num_buckets > 1:
        # buckets tile the arena exactly, last-finished layers first
        rs = step.bucket_ranges
        assert rs[-1][0] == 0 and rs[0][1] == step.arena.total
        for (lo, hi), (lo2, hi2) in zip(rs[:-1], rs[1:]):
            assert hi2 == lo and lo2 < hi2
        assert step.num_segments == len(rs) + 1


@pytest.mark.parametrize("optimizer", ["adam", "momentum"])
def test_fused_update_and_graph_replay(optimizer):
    n, batch = 1, 8
    step, images, labels = _make(n, batch, optimizer=optimizer)
    _, grads_ref, params = _reference_grads(step, images, labels, n, torch.float32)
    lr_t = ref.adam_lr_t(1e-3, 1) if optimizer == "adam" else 0.05
    want = {}
    for name, p in params.items():
        p0 = p.detach().clone()
        g = grads_ref[name].float() * 0.5
        if optimizer == "adam":
            m, v = torch.zeros_like(p0), torch.zeros_like(p0)
            ref.adam_update(p0, g, m, v, lr_t)
        else:
            m = torch.zeros_like(p0)
            ref.momentum_update(p0, g, m, lr_t, 0.9, 2e-4)
        want[name] = p0
    step.set_hyper(lr_t=lr_t, grad_scale=0.5, momentum=0.9, weight_decay=2e-4)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step.instantiate()
        step.run(graph=True)
    side.synchronize()
    for name, w in want.items():
        got = step.arena.param(name).cpu()
        # Adam's first step is lr*sign(g): compare where the reference gradient is not tiny
        if optimizer == "adam":
            mask = grads_ref[name].abs() > 1e-6
            assert (got - w)[mask].abs().max() < 3e-4 if mask.any() else True
        else:
            assert _rel(got, w) < 1e-4, name
    # the loss goes down when the same batch is replayed
    with torch.cuda.stream(side):
        l0 = step.loss_value()
        for _ in range(20):
            step.run(graph=True)
        side.synchronize()
        l1 = step.loss_value()
    assert l1 < l0


def test_eval_program_counts_top1():
    from distributed_tensorflow_amd.models.resnet_hip import ResNetHipStep
    from distributed_tensorflow_amd.models.resnet import resnet_forward

    n, batch = 1, 16
    arena = ParamArena(resnet_param_specs(n), "cuda")
    arena.initialize()
    ev = ResNetHipStep(n, batch, arena, training=False)
    g = torch.Generator().manual_seed(5)
    images = torch.randn(batch, 32, 32, 3, generator=g)
    labels = torch.randint(0, 10, (batch,), generator=g)
    ev.load_batch(images.cuda(), labels.cuda())
    ev.run(graph=False)
    torch.cuda.synchronize()
    logits = resnet_forward(arena.state_dict(), images, n)
    assert ev.correct_count() == ref.top1_hits(logits, labels)
    assert abs(ev.loss_value() - ref.softmax_xent_mean(logits, labels).item()) < 1e-3


def test_gpu_batcher_matches_host_batcher():
    import numpy as np

    from distributed_tensorflow_amd.data import cifar10 as cd

    data, labels = cd.synthetic_cifar(96, 5)
    padded = cd.pad_images(data, 2)
    host = cd.TrainBatcher(padded.copy(), labels.copy(), 2, seed=11)
    gpu = cd.GpuTrainBatcher(padded, labels, 2, seed=11)
    images = torch.empty(40, 32, 32, 3, device="cuda")
    lab = torch.empty(40, device="cuda", dtype=torch.int32)
    for _ in range(4):   # crosses an epoch boundary (reshuffle)
        want_x, want_y = host.next_batch(40)
        gpu.next_batch_into(images, lab)
        torch.cuda.synchronize()
        assert np.array_equal(lab.cpu().numpy(), want_y)
        assert np.abs(images.cpu().numpy() - want_x).max() < 2e-4


@pytest.mark.parametrize("env", [dict(DTF_BWD_FUSED="0", DTF_SIDE_STREAM="1"), dict(DTF_BWD_FUSED="0", DTF_SIDE_STREAM="0"),
                                 dict(DTF_BWD_FUSED="1", DTF_SIDE_STREAM="1"),
                                 dict(DTF_BWD_FUSED="1", DTF_SIDE_STREAM="0", DTF_BWD_SHORTCUT_FUSE="", DTF_BWD_STEM_FOLD="0"),
                                 dict(DTF_BWD_FUSED="1", DTF_SIDE_STREAM="0", DTF_BWD_SHORTCUT_FUSE="16,32,64")])
def test_backward_variants_agree(monkeypatch, env):
    """Separate data/filter-gradient kernels (with the filter gradients on a side lane or not), the
    fused backward kernels, and the fused kernels with the identity blocks' input gradient folded
    into the consuming convolution (switched off / on) must
    produce the same gradients and the same update."""
    from distributed_tensorflow_amd.models.resnet_hip import ResNetHipStep

    b = 16
    g = torch.Generator().manual_seed(9)
    images = torch.randn(b, 32, 32, 3, generator=g).cuda()
    labels = torch.randint(0, 10, (b,), generator=g).cuda()

    def grads_with(envd):
        for k, v in envd.items():
            monkeypatch.setenv(k, v)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            step = ResNetHipStep(3, b, num_buckets=2)
            step.load_batch(images, labels)
            step.set_hyper(lr_t=0.0)
            step.run(graph=False)
            s.synchronize()
            out = step.arena.grads.clone(), step.loss_value()
            step.instantiate(per_segment=True)
            for i in range(step.num_segments):
                step.run(i, True)
            s.synchronize()
            again = step.arena.grads.clone()
        return out[0], out[1], again

    base, loss0, _ = grads_with(dict(DTF_BWD_FUSED="1", DTF_SIDE_STREAM="0"))
    got, loss1, replay = grads_with(env)
    assert abs(loss0 - loss1) < 1e-5 * max(1.0, abs(loss0))
    scale = base.abs().max().item()
    assert (got - base).abs().max().item() < 2e-4 * scale
    assert (replay - got).abs().max().item() < 2e-4 * scale   # hipGraph replay of the forked program


def test_bench_contract_one_json_line_through_the_rccl_path():
    """bench.py with a one-rank RCCL process group: the bucketed step issues its all-reduces
    through the real backend, and stdout carries exactly one JSON line (RCCL's banner and any
    other library chatter go to stderr)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29713", NCCL_DEBUG="VERSION", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for extra in ([], ["--mode", "sharded"]):
        r = subprocess.run([sys.executable, "bench.py", "--rehearse-collectives", "--model", "resnet8", "--batch", "32",
                            "--steps", "3", "--warmup", "1"] + extra, capture_output=True, text=True, timeout=300, cwd=root, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [l for l in r.stdout.splitlines() if l.strip()]
        assert len(lines) == 1, r.stdout
        d = json.loads(lines[0])
        assert d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 0 and d["dtype"] == "fp32"
        assert d["config"]["final_loss"] == d["config"]["final_loss"]   # not NaN


def test_graph_capture_next_to_pending_collectives():
    """The process group's watchdog thread polls the end events of collectives issued on the
    compute stream; on this ROCm an event query fails, and ends the process, while the event's
    stream is capturing (bench/micro/event_query_during_capture.py). Step programs are therefore
    captured on a stream of their own: capture right behind a collective, many times, with a
    one-rank RCCL group."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import torch, torch.distributed as dist
from distributed_tensorflow_amd.models.resnet_hip import ResNetHipStep
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    step = ResNetHipStep(1, 8, num_buckets=3)
    t = torch.ones(1 << 16, device="cuda")
    for i in range(150):
        dist.all_reduce(t)                       # end event recorded on the current stream
        step.instantiate(per_segment=True)       # capture right behind it
        for seg in range(step.num_segments):
            step.run(seg, True)
    s.synchronize()
dist.destroy_process_group()
print("ok")
"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29717", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-3000:]


def test_queued_steps_keep_their_own_scalars():
    """Steps queued without any host synchronisation must each run with the learning rate and
    gradient scale staged for them (the host runs far ahead of the GPU in the benchmark loop):
    same parameters, bit for bit, as a run that synchronises after every step. Whole steps -
    forward, backward, optimizer - which is only a meaningful comparison because the step is
    deterministic (per-workgroup gradient copies summed in a fixed order, statistics accumulated
    exactly): Adam would amplify any last-bit difference between two runs."""
    n, batch, steps = 1, 64, 24

    def run(sync_each, shift=0):
        step, _, _ = _make(n, batch)
        assert step.deterministic
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            step.instantiate(per_segment=True)
            for t in range(1, steps + 1):
                # values that differ a lot from step to step, so a mix-up is visible
                k = t + shift
                step.set_hyper(lr_t=ref.adam_lr_t(1e-3, t) * (1.0 + (k % 3)), grad_scale=1.0 / (1 + k % 4))
                for seg in range(step.num_segments):
                    step.run(seg, True)
                if sync_each:
                    s.synchronize()
        s.synchronize()
        return step.arena.params.clone()

    synced = run(True)
    queued = run(False)
    mixed = run(True, shift=1)
    assert not torch.equal(synced, mixed)        # the yardstick: scalars one step off are visible
    assert torch.equal(synced, queued)


@pytest.mark.parametrize("matmul", ["fp32", "bf16x6"])
@pytest.mark.parametrize("n,batch,num_buckets", [(1, 64, 1), (1, 256, 3), (3, 128, 1)])
def test_training_is_reproducible_bit_for_bit(n, batch, num_buckets, matmul):
    """Two runs of 24 training steps from the same state end with identical parameters, optimizer
    slots and loss-carrying gradients: no float atomics in arrival order anywhere in the step
    (filter gradients: one copy per workgroup, added in a fixed order; statistics: exact fp64
    accumulation on a grid; classifier gradients: fixed-order in-block sums)."""
    def run():
        step, _, _ = _make(n, batch, num_buckets=num_buckets, matmul=matmul)
        assert step.deterministic
        if matmul == "bf16x6":
            assert step._split_fwd and step._split_bwd   # both split-operand kernel families are in the step
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            step.instantiate(per_segment=True)
            for t in range(1, 25):
                step.set_hyper(lr_t=ref.adam_lr_t(1e-3, t))
                for seg in range(step.num_segments):
                    step.run(seg, True)
        s.synchronize()
        return step.arena.params.clone(), step.arena.grads.clone(), step.arena.slots["m"].clone(), step.arena.slots["v"].clone()

    a, b = run(), run()
    for x, y, what in zip(a, b, ("parameters", "gradients", "m", "v")):
        assert torch.equal(x, y), f"{what} differ: max |d| = {float((x - y).abs().max()):.3e}"


def test_atomic_and_copy_gradient_flush_agree():
    """DTF_DETERMINISTIC=0 keeps the float-atomic flush (what short-circuiting replicas run): same
    gradients up to summation order."""
    step_d, images, labels = _make(3, 64)
    os.environ["DTF_DETERMINISTIC"] = "0"
    try:
        step_a, _, _ = _make(3, 64)
    finally:
        del os.environ["DTF_DETERMINISTIC"]
    assert step_d.deterministic and not step_a.deterministic
    for st in (step_d, step_a):
        st.set_hyper(lr_t=0.0)
        st.run(graph=False)
    torch.cuda.synchronize()
    for name in step_d.arena.names():
        assert _rel(step_a.arena.grad(name), step_d.arena.grad(name)) < 2e-5, name


def test_set_scalars_values_travel_with_the_launch():
    from distributed_tensorflow_amd.ops import native as nv

    f = torch.zeros(8, device="cuda")
    u = torch.zeros(2, device="cuda", dtype=torch.int64)
    outs = []
    for k in range(50):
        nv.set_scalars(f, (k, 2 * k, 0.5), u, (k << 20,))
        outs.append((f.clone(), u.clone()))          # stream-ordered snapshots, no host sync in between
    torch.cuda.synchronize()
    for k, (fo, uo) in enumerate(outs):
        assert fo[:3].tolist() == [float(k), float(2 * k), 0.5] and int(uo[0]) == k << 20


def test_traced_step_timeline_has_one_event_per_kernel_and_the_phases(tmp_path):
    """A traced ResNet-20 step: every launch of the step program shows up as a named event with a
    duration, on the same time axis as the phase spans (the per-op chrome trace of the reference's
    --timeline_logging, DistributedLeNet/src/distributed_train.py:300-328)."""
    import json

    from distributed_tensorflow_amd.parallel.comm import Communicator
    from distributed_tensorflow_amd.parallel.strategies import FullSync
    from distributed_tensorflow_amd.train.replica import HipReplica
    from distributed_tensorflow_amd.train.timeline import StepTimeline

    rep = HipReplica("resnet", 64, num_residual_blocks=3)
    g = torch.Generator().manual_seed(3)
    rep.load_batch(torch.randn(64, 32, 32, 3, generator=g), torch.randint(0, 10, (64,), generator=g))
    strat = FullSync(Communicator.single("cuda"))
    strat.step(rep, 0, 1e-3)                       # an ordinary (graph-replayed) step first
    rep.synchronize()
    before = rep.arena.params.clone()
    tl = StepTimeline(0, enabled=True)
    rep.timeline = tl
    tl.reset(rep.stream)
    with tl.span("train_step", rep.stream):
        strat.step(rep, 1, 1e-3)
    rep.synchronize()
    assert not torch.equal(before, rep.arena.params)          # the traced step is a real step
    path = tl.write(str(tmp_path), 1)
    events = json.load(open(path))["traceEvents"]
    kernels = [e for e in events if e.get("cat") == "kernel"]
    assert len(kernels) == rep.step.kernel_count >= 50
    assert all(e["dur"] > 0 and e["ts"] >= 0 and e["args"]["blocks"] >= 0 for e in kernels)
    names = {e["name"] for e in kernels}
    assert any("conv" in n for n in names) and any("adam" in n for n in names), names
    phases = {e["name"]: e for e in events if e.get("cat") == "phase"}
    assert {"prologue", "forward", "backward", "optimizer"} <= set(phases)
    assert phases["forward"]["ts"] < phases["backward"]["ts"] < phases["optimizer"]["ts"]
    spans = {e["name"] for e in events if e.get("cat") == "gpu"}
    assert {"train_step", "forward+backward", "optimizer"} <= spans
    # kernels lie inside the step's GPU span
    step = next(e for e in events if e.get("cat") == "gpu" and e["name"] == "train_step")
    assert all(step["ts"] - 50 <= e["ts"] and e["ts"] + e["dur"] <= step["ts"] + step["dur"] + 50 for e in kernels)
    # back to graph replay afterwards
    rep.timeline = None
    strat.step(rep, 2, 1e-3)
    rep.synchronize()
