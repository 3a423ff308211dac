"""Process flags with the `tf.app.flags` surface of the reference.

The reference defines its flags piecemeal across the training drivers, optimizers, evaluators and
the ResNet hyper-parameter file (reference: DistributedLeNet/src/distributed_train.py:39-98,
DistributedResNet/src/hyper_parameters.py:11-46, */src/nn_eval.py:36-45,
soft_kill_optimizer.py:43, sync_replicas_optimizer_modified.py:39,
Distributed_test/src/distributed_train.py:59-60). Here the complete set is registered in one
place with the same names and defaults, so command lines written for the reference parse
unchanged: `--name=value`, `--name value`, bare `--bool_flag`, `--nobool_flag`, and the
`true/false/True/False/1/0` spellings the reference's cfg files use.

Flags that only make sense on this framework (data source, optimizer choice, fault injection,
async staleness bound, ...) are registered alongside and documented as extensions.
"""
from __future__ import annotations

import sys
from typing import Any, Callable, Dict, List, Optional, Sequence


class _Flag:
    __slots__ = ("name", "default", "help", "parser", "kind", "value", "present")

    def __init__(self, name: str, default: Any, help_: str, parser: Callable[[str], Any], kind: str):
        self.name, self.default, self.help, self.parser, self.kind = name, default, help_, parser, kind
        self.value = default
        self.present = False


def _parse_bool(s: str) -> bool:
    v = s.strip().lower()
    if v in ("true", "t", "1", "yes", "y"):
        return True
    if v in ("false", "f", "0", "no", "n"):
        return False
    raise ValueError(f"not a boolean: {s!r}")


class FlagValues:
    """Registry + attribute access (`FLAGS.batch_size`), parsed lazily from sys.argv like TF."""

    def __init__(self) -> None:
        object.__setattr__(self, "_flags", {})
        object.__setattr__(self, "_parsed", False)
        object.__setattr__(self, "_unparsed", [])

    # -- registration ----------------------------------------------------------------------------
    def _define(self, name: str, default: Any, help_: str, parser: Callable[[str], Any], kind: str) -> None:
        flags: Dict[str, _Flag] = self._flags
        if name in flags:
            old = flags[name]
            if old.kind != kind:
                raise ValueError(f"flag {name} redefined with a different type")
            return  # the reference re-defines flags across modules; first definition wins
        flags[name] = _Flag(name, default, help_, parser, kind)

    # -- parsing -----------------------------------------------------------------------------------
    def parse(self, argv: Optional[Sequence[str]] = None, known_only: bool = True) -> List[str]:
        """Parses `argv` (without the program name); returns the arguments that are not flags."""
        argv = list(sys.argv[1:] if argv is None else argv)
        flags: Dict[str, _Flag] = self._flags
        rest: List[str] = []
        i = 0
        while i < len(argv):
            arg = argv[i]
            i += 1
            if arg == "--":
                rest.extend(argv[i:])
                break
            if not arg.startswith("-") or arg in ("-", ):
                rest.append(arg)
                continue
            body = arg.lstrip("-")
            if "=" in body:
                name, val = body.split("=", 1)
                has_val = True
            else:
                name, val, has_val = body, "", False
            f = flags.get(name)
            if f is None and name.startswith("no") and name[2:] in flags and flags[name[2:]].kind == "bool" and not has_val:
                f = flags[name[2:]]
                f.value, f.present = False, True
                continue
            if f is None:
                if known_only:
                    rest.append(arg)
                    continue
                raise ValueError(f"unknown flag --{name}")
            if f.kind == "bool":
                if has_val:
                    f.value = _parse_bool(val)
                elif i < len(argv) and argv[i].strip().lower() in ("true", "false"):
                    f.value = _parse_bool(argv[i])
                    i += 1
                else:
                    f.value = True
            else:
                if not has_val:
                    if i >= len(argv):
                        raise ValueError(f"flag --{name} needs a value")
                    val = argv[i]
                    i += 1
                f.value = f.parser(val)
            f.present = True
        object.__setattr__(self, "_parsed", True)
        object.__setattr__(self, "_unparsed", rest)
        return rest

    def reset(self) -> None:
        for f in self._flags.values():
            f.value, f.present = f.default, False
        object.__setattr__(self, "_parsed", False)

    # -- access ------------------------------------------------------------------------------------
    def __getattr__(self, name: str) -> Any:
        flags: Dict[str, _Flag] = object.__getattribute__(self, "_flags")
        if name not in flags:
            raise AttributeError(f"no flag named {name}")
        if not object.__getattribute__(self, "_parsed"):
            self.parse(known_only=True)
        return flags[name].value

    def __setattr__(self, name: str, value: Any) -> None:
        flags: Dict[str, _Flag] = self._flags
        if name not in flags:
            raise AttributeError(f"no flag named {name}")
        flags[name].value = value
        flags[name].present = True

    def is_present(self, name: str) -> bool:
        return self._flags[name].present

    def flag_values_dict(self) -> Dict[str, Any]:
        if not self._parsed:
            self.parse(known_only=True)
        return {k: f.value for k, f in self._flags.items()}

    def help_text(self) -> str:
        return "\n".join(f"  --{f.name} ({f.kind}, default {f.default!r}): {f.help}" for f in self._flags.values())


FLAGS = FlagValues()


def DEFINE_boolean(name: str, default: bool, help: str = "") -> None:  # noqa: N802,A002
    FLAGS._define(name, default, help, _parse_bool, "bool")


DEFINE_bool = DEFINE_boolean


def DEFINE_string(name: str, default: Optional[str], help: str = "") -> None:  # noqa: N802,A002
    FLAGS._define(name, default, help, str, "string")


def DEFINE_integer(name: str, default: int, help: str = "") -> None:  # noqa: N802,A002
    FLAGS._define(name, default, help, lambda s: int(float(s)) if "." in s or "e" in s.lower() else int(s), "int")


def DEFINE_float(name: str, default: float, help: str = "") -> None:  # noqa: N802,A002
    FLAGS._define(name, default, help, float, "float")


def define_reference_flags() -> None:
    """Every flag of the reference, with the reference's defaults."""
    # --- synchronisation strategy switches (distributed_train.py:39-44)
    DEFINE_boolean("worker_times_cdf_method", False, "Track worker times cdf")
    DEFINE_boolean("interval_method", False, "Use the interval method")
    DEFINE_boolean("backup_worker_method", False, "Use for backup worker experiments")
    DEFINE_boolean("soft_kill_method", False, "Use for soft kill experiments")
    DEFINE_boolean("should_summarize", False, "Whether Chief should write summaries.")
    DEFINE_boolean("timeline_logging", False, "Whether to log timeline of events.")
    # --- cluster (distributed_train.py:45-59)
    DEFINE_string("job_name", "", 'One of "ps", "worker"')
    DEFINE_string("ps_hosts", "", "Comma-separated list of hostname:port for the parameter server jobs.")
    DEFINE_string("worker_hosts", "", "Comma-separated list of hostname:port for the worker jobs.")
    DEFINE_string("train_dir", "/tmp/imagenet_train", "Directory where to write event logs and checkpoint.")
    DEFINE_integer("rpc_port", 1235, "Port for timeout communication")
    # --- loop (distributed_train.py:61-80)
    DEFINE_integer("max_steps", 1000000, "Number of batches to run.")
    DEFINE_integer("batch_size", 128, "Batch size.")
    DEFINE_string("subset", "train", 'Either "train" or "validation".')
    DEFINE_boolean("log_device_placement", False, "Whether to log device placement.")
    DEFINE_integer("task_id", 0, "Task ID of the worker/replica running the training.")
    DEFINE_integer("num_replicas_to_aggregate", -1, "Number of gradients to collect before updating the parameters.")
    DEFINE_integer("save_interval_secs", 20, "Save interval seconds.")
    DEFINE_integer("save_summaries_secs", 300, "Save summaries interval seconds.")
    # --- learning rate schedule (distributed_train.py:93-98)
    DEFINE_float("initial_learning_rate", 0.1, "Initial learning rate.")
    DEFINE_float("num_epochs_per_decay", 2.0, "Epochs after which learning rate decays.")
    DEFINE_float("learning_rate_decay_factor", 0.999, "Learning rate decay factor.")
    # --- strategy parameters
    DEFINE_integer("interval_ms", 1000, "The interval ms")                          # sync_replicas_optimizer_modified.py:39
    DEFINE_integer("should_stop_worker_num", 2, "Gradients after which stragglers are soft-killed")  # soft_kill_optimizer.py:43
    DEFINE_integer("num_worker_kill", 3, "Number of workers to kill at each iteration")              # Distributed_test/src/distributed_train.py:59
    DEFINE_integer("random_seed", 142, "Random seed for the worker-kill experiment")                 # Distributed_test/src/distributed_train.py:60
    DEFINE_integer("num_of_instances_cifar10", 50000, "Number of CIFAR-10 training instances")       # DistributedResNet/src/distributed_train.py
    # --- evaluation (nn_eval.py:36-45)
    DEFINE_string("eval_dir", "/tmp/imagenet_eval", "Directory where to write event logs.")
    DEFINE_string("checkpoint_dir", "/tmp/imagenet_train", "Directory where to read model checkpoints.")
    DEFINE_integer("eval_interval_secs", 1, "How often to run the eval.")
    DEFINE_boolean("run_once", False, "Whether to run eval only once.")
    # --- ResNet hyper-parameters (hyper_parameters.py:11-46)
    DEFINE_string("version", "test_110", "A version number defining the directory to save logs and checkpoints")
    DEFINE_integer("report_freq", 391, "Steps takes to output errors on the screen and write summaries")
    DEFINE_float("train_ema_decay", 0.95, "The decay factor of the train error's moving average shown on tensorboard")
    DEFINE_integer("train_steps", 80000, "Total steps that you want to train")
    DEFINE_boolean("is_full_validation", True, "Validation w/ full validation set or a random batch")
    DEFINE_integer("train_batch_size", 128, "Train batch size")
    DEFINE_integer("validation_batch_size", 250, "Validation batch size, better to be a divisor of 10000")
    DEFINE_integer("test_batch_size", 125, "Test batch size")
    DEFINE_float("init_lr", 0.1, "Initial learning rate")
    DEFINE_float("lr_decay_factor", 0.1, "How much to decay the learning rate each time")
    DEFINE_integer("decay_step0", 40000, "At which step to decay the learning rate")
    DEFINE_integer("decay_step1", 60000, "At which step to decay the learning rate")
    DEFINE_integer("num_residual_blocks", 18, "How many residual blocks do you want")
    DEFINE_float("weight_decay", 0.0002, "scale for l2 regularization")
    DEFINE_integer("padding_size", 2, "In data augmentation, layers of zero padding on each side of the image")
    DEFINE_string("ckpt_path", "cache/logs_repeat20/model.ckpt-100000", "Checkpoint directory to restore")
    DEFINE_boolean("is_use_ckpt", False, "Whether to load a checkpoint and continue training")
    DEFINE_string("test_ckpt_path", "model_110.ckpt-79999", "Checkpoint directory to restore")


def define_extension_flags() -> None:
    """Flags that exist only in this framework."""
    DEFINE_string("model", "", "Workload: lenet, resnet (set by the entry scripts)")
    DEFINE_string("data_dir", "", "Directory holding MNIST idx / CIFAR-10 python-batch files; empty = synthetic data")
    DEFINE_boolean("synthetic_data", False, "Use synthetic data of the dataset's shape (no files needed)")
    DEFINE_string("optimizer", "adam", "adam (the reference's choice) or momentum (momentum SGD + weight decay)")
    DEFINE_string("matmul", "fp32", "ResNet on the GPU: fp32 = every product on the exact-fp32 matrix instruction; bf16x6 = the stride-1 "
                  "3x3 layers on the bf16 matrix pipe with fp32 operands split three ways and fp32 accumulation (opt-in)")
    DEFINE_float("momentum", 0.9, "Momentum of the momentum-SGD optimizer")
    DEFINE_string("device", "auto", "auto, cuda or cpu")
    DEFINE_boolean("async_method", False, "Asynchronous SGD over staleness-bounded parameter shards")
    DEFINE_integer("interval_max_staleness", -1, "Interval mode: drop a gradient computed on parameters more than this many "
                   "updates old (-1: accept any, the gradient counts for the window that is open when it is pushed, like the reference)")
    DEFINE_integer("staleness_bound", -1, "Async mode: gradients older than this many shard versions are dropped "
                   "(0: only fresh ones, the accumulator's rule; -1: twice the number of workers - every worker "
                   "pushes to every shard once per iteration, so a version is a round old after `workers` pushes)")
    DEFINE_boolean("weighted_grads_method", False, "Least-squares weighted gradient aggregation with simulated worker kills")
    DEFINE_integer("grad_buckets", 0, "Gradient all-reduce buckets overlapped with the backward pass (0: three for the "
                   "straggler policies, which poll at bucket boundaries, else by gradient size)")
    DEFINE_boolean("reduce_scatter", False, "Shard the optimizer: reduce-scatter gradients, all-gather parameters")
    DEFINE_boolean("gpu_data_pipeline", True, "On a GPU keep the training set in HBM (ResNet: crop + whiten on the device; LeNet: batches sliced device-to-device)")
    DEFINE_string("allreduce_impl", "rccl", "Gradient all-reduce: rccl (RCCL collectives) or p2p (hand-written two-shot kernel over peer-mapped HBM)")
    DEFINE_boolean("hip_graph", True, "Replay the training step as a hipGraph")
    DEFINE_integer("inject_straggler_ms", 0, "Fault injection: delay the listed replicas by this many ms per step")
    DEFINE_string("inject_straggler_ranks", "", "Comma-separated replica ids to delay")
    DEFINE_float("inject_straggler_prob", 1.0, "Probability that a listed replica is delayed in a given step")
    DEFINE_integer("log_every", 1, "Log the training line every this many steps (1 = every step, like the reference)")
    DEFINE_string("checkpoint_names", "tf_auto", "Variable naming in checkpoints: tf_auto (Variable_N) or scoped")
    DEFINE_string("board_name", "", "Shared-memory control board name (set by the launcher)")
    DEFINE_integer("collective_timeout_s", 600, "Watchdog timeout of collectives and barriers")
    DEFINE_string("comm_backend", "", "torch.distributed backend (default: nccl = RCCL for GPU replicas, gloo for CPU); "
                  "gloo lets several GPU replicas share one device for rehearsals")
    DEFINE_string("control_plane", "auto", "Where the workers coordinate: shm (shared-memory board, one node), store (the process "
                  "group's key-value store: workers on several machines, gradients of the straggler policies travel through "
                  "it), auto (store when the ranks span machines)")
    DEFINE_float("liveness_timeout_s", 0.0, "A worker silent for this long is declared dead (0: collective_timeout_s / 4)")
    DEFINE_integer("eval_batch_size", 0, "Evaluator batch size; 0 = the whole validation set as one batch, like the reference")
    DEFINE_integer("max_evals", 0, "Evaluator: stop after this many evaluated checkpoints (0 = unbounded)")


define_reference_flags()
define_extension_flags()
