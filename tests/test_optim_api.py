"""Public optimizer API: flat-arena Adam / momentum SGD against a straight PyTorch re-statement of
the update rules, and the reference-named synchronisation wrappers on a single replica."""
import numpy as np
import pytest
import torch

from distributed_tensorflow_amd.models.lenet import lenet_param_specs
from distributed_tensorflow_amd.optim import (BackupOptimizer, FlatAdam, FlatMomentumSGD, SoftKillOptimizer,
                                              SyncReplicasOptimizer, TimeoutReplicasOptimizer, WeightedGradsOptimizer)
from distributed_tensorflow_amd.parallel.comm import Communicator
from distributed_tensorflow_amd.runtime.arena import ParamArena
from distributed_tensorflow_amd.train.replica import make_replica


def _arena(slots, device="cpu"):
    a = ParamArena(lenet_param_specs(), device, slots)
    a.initialize(1)
    return a


def test_flat_adam_matches_tf_adam_rule():
    """tf.train.AdamOptimizer: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); p -= lr_t*m/(sqrt(v)+eps). (torch's
    Adam puts epsilon inside the bias correction, which differs where a gradient is ~0.)"""
    a = _arena(("m", "v"))
    p = a.params.clone().double()
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    opt = FlatAdam(a)
    g = torch.Generator().manual_seed(0)
    for t in range(1, 5):
        grad = torch.randn(a.total, generator=g)
        a.grads.copy_(grad * 2.0)
        opt.step(1e-3, grad_scale=0.5)
        gd = grad.double()
        m = 0.9 * m + 0.1 * gd
        v = 0.999 * v + 0.001 * gd * gd
        p -= 1e-3 * (1 - 0.999 ** t) ** 0.5 / (1 - 0.9 ** t) * m / (v.sqrt() + 1e-8)
    assert opt.steps == 4
    assert torch.allclose(a.params.double(), p, rtol=1e-5, atol=1e-7)


def test_flat_adam_range_leaves_rest_untouched():
    a = _arena(("m", "v"))
    p0 = a.params.clone()
    a.grads.fill_(1.0)
    FlatAdam(a).step(1e-2, lo=100, hi=200)
    assert torch.equal(a.params[:100], p0[:100]) and torch.equal(a.params[200:], p0[200:])
    assert not torch.equal(a.params[100:200], p0[100:200])
    with pytest.raises(ValueError):
        FlatAdam(a).step(1e-2, lo=10, hi=a.total + 1)


def test_flat_momentum_sgd_weight_decay():
    a = _arena(("m",))
    ref_p = torch.nn.Parameter(a.params.clone())
    ref = torch.optim.SGD([ref_p], lr=0.1, momentum=0.9, weight_decay=2e-4)
    opt = FlatMomentumSGD(a, momentum=0.9, weight_decay=2e-4)
    g = torch.Generator().manual_seed(3)
    for _ in range(3):
        grad = torch.randn(a.total, generator=g)
        a.grads.copy_(grad)
        opt.step(0.1)
        ref_p.grad = grad.clone()
        ref.step()
    assert torch.allclose(a.params, ref_p.detach(), rtol=1e-5, atol=1e-6)


def test_missing_slot_is_an_error():
    with pytest.raises(ValueError):
        FlatAdam(_arena(("m",)))


@pytest.mark.parametrize("make", [
    lambda c: SyncReplicasOptimizer("adam", comm=c),
    lambda c: BackupOptimizer("adam", replicas_to_aggregate=1, total_num_replicas=1, comm=c),
    lambda c: TimeoutReplicasOptimizer("adam", 0, comm=c, interval_ms=60000.0),
    lambda c: SoftKillOptimizer("adam", 0, comm=c, should_stop_worker_num=1),
    lambda c: WeightedGradsOptimizer("adam", comm=c, num_worker_kill=0),
])
def test_reference_named_wrappers_single_replica(make):
    comm = Communicator.single("cpu")
    opt = make(comm)
    rep = make_replica("lenet", 8, device="cpu")
    rng = np.random.RandomState(0)
    x = rng.rand(8, 28, 28, 1).astype(np.float32) - 0.5
    y = rng.randint(0, 10, size=8)
    before = rep.arena.params.clone()
    rep.load_batch(x, y)
    res = opt.step(rep, 0, 1e-3)
    assert res.applied and res.contributors == 1
    assert not torch.equal(before, rep.arena.params)


def test_two_phase_api_matches_one_call_step():
    """compute_gradients + apply_gradients(grads, worker_id, global_step) is the same step as .step()."""
    comm = Communicator.single("cpu")
    rng = np.random.RandomState(0)
    x = rng.rand(8, 28, 28, 1).astype(np.float32) - 0.5
    y = rng.randint(0, 10, size=8)
    reps = [make_replica("lenet", 8, device="cpu") for _ in range(2)]
    for r in reps:
        r.load_batch(x, y)
    one = BackupOptimizer("adam", 1, comm=comm)
    one.step(reps[0], 0, 1e-3)
    two = BackupOptimizer("adam", 1, comm=comm)
    runner = two.get_chief_queue_runner().start()
    two.get_init_tokens_op()()
    two.chief_init_op(), two.local_step_init_op()
    assert two.ready_for_local_init_op()
    grads = two.compute_gradients(reps[1])
    assert float(reps[1].arena.grads.abs().sum()) > 0          # the gradient is there before anything is applied
    res = two.apply_gradients(grads, worker_id=0, global_step=0, lr=1e-3)
    assert res.applied and res.global_step == 1
    assert torch.equal(reps[0].arena.params, reps[1].arena.params)
    with pytest.raises(ValueError):
        two.apply_gradients(two.compute_gradients(reps[1]), worker_id=3, global_step=1, lr=1e-3)
    runner.stop()
    hook = WeightedGradsOptimizer("adam", comm=comm, num_worker_kill=0).make_session_run_hook(True)
    hook.begin()
    hook.after_create_session()
    assert hook.runner.started


def test_wrapper_rejects_wrong_replica_count():
    with pytest.raises(ValueError):
        BackupOptimizer("adam", 2, total_num_replicas=4, comm=Communicator.single("cpu"))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["adam", "momentum"])
def test_flat_optimizers_gpu_match_cpu(kind):
    slots = ("m", "v") if kind == "adam" else ("m",)
    cpu, gpu = _arena(slots), _arena(slots, "cuda")
    gpu.params.copy_(cpu.params)
    mk = (lambda a: FlatAdam(a)) if kind == "adam" else (lambda a: FlatMomentumSGD(a, 0.9, 2e-4))
    oc, og = mk(cpu), mk(gpu)
    g = torch.Generator().manual_seed(5)
    lo, hi = 1024, cpu.total - 4096
    for i in range(3):
        grad = torch.randn(cpu.total, generator=g)
        cpu.grads.copy_(grad)
        gpu.grads.copy_(grad)
        if i == 2:   # a sub-range, as the sharded optimizer uses it
            oc.step(1e-3, 0.25, lo, hi)
            og.step(1e-3, 0.25, lo, hi)
        else:
            oc.step(1e-3, 0.25)
            og.step(1e-3, 0.25)
    torch.cuda.synchronize()
    assert torch.allclose(gpu.params.cpu(), cpu.params, rtol=1e-5, atol=1e-7)
